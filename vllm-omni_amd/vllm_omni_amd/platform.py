"""MI355X platform plugin: the drop-in boundary on the reference's side.

An out-of-tree platform for vLLM-Omni is a Python entry point in group ``vllm_omni.platform_plugins``
whose callable returns the platform class qualname or ``None``
(vllm_omni/plugins/__init__.py:18,24-58; vllm_omni/platforms/__init__.py:133-165; an OOT plugin wins
over the built-in ROCm platform, :151-155).  ``MI355XOmniPlatform`` implements exactly the
``OmniPlatform`` method set (vllm_omni/platforms/interface.py:26-165), modelled on
``RocmOmniPlatform`` (vllm_omni/platforms/rocm/platform.py:49-143), and names THIS package's worker
as the AR worker class.

When vllm / vllm_omni are importable the class subclasses the real ``OmniPlatform`` (and upstream's
``RocmPlatform``); in a container without them (this build container) it subclasses a local stand-in
with the same abstract surface so the plugin stays importable and testable.
"""
from __future__ import annotations

import os
from contextlib import nullcontext
from enum import Enum
from typing import Optional

import torch

try:  # real bases when the reference stack is installed
    from vllm.platforms.rocm import RocmPlatform as _UpstreamBase  # type: ignore
    from vllm_omni.platforms.interface import OmniPlatform as _OmniBase, OmniPlatformEnum  # type: ignore

    HAVE_REFERENCE_STACK = True
except Exception:  # noqa: BLE001 - any import problem means "not installed here"
    HAVE_REFERENCE_STACK = False

    class OmniPlatformEnum(Enum):  # mirrors platforms/interface.py:15-23
        CUDA = "cuda"
        ROCM = "rocm"
        NPU = "npu"
        XPU = "xpu"
        MUSA = "musa"
        UNSPECIFIED = "unspecified"

    class _UpstreamBase:  # the attributes of upstream ``Platform`` the reference reads
        device_name = "rocm"
        device_type = "cuda"
        dispatch_key = "CUDA"
        dist_backend = "nccl"  # == RCCL on ROCm (vllm_omni/worker/gpu_ar_worker.py:74)
        device_control_env_var = "CUDA_VISIBLE_DEVICES"

        @classmethod
        def check_if_supports_dtype(cls, dtype: torch.dtype) -> None:
            return None

    class _OmniBase:  # abstract surface of OmniPlatform (interface.py:26-165)
        _omni_enum: OmniPlatformEnum

        def is_npu(self): return self._omni_enum == OmniPlatformEnum.NPU
        def is_xpu(self): return self._omni_enum == OmniPlatformEnum.XPU
        def is_cuda(self): return self._omni_enum == OmniPlatformEnum.CUDA
        def is_rocm(self): return self._omni_enum == OmniPlatformEnum.ROCM
        def is_musa(self): return self._omni_enum == OmniPlatformEnum.MUSA


PLATFORM_QUALNAME = "vllm_omni_amd.platform.MI355XOmniPlatform"
AR_WORKER_QUALNAME = "vllm_omni_amd.worker.GPUARWorker"
SUPPORTED_ARCHS = ("gfx950",)


class MI355XOmniPlatform(_OmniBase, _UpstreamBase):
    """AMD Instinct MI355X (gfx950 / CDNA4) platform: HIP kernels behind a C ABI, RCCL over xGMI."""

    _omni_enum = OmniPlatformEnum.ROCM
    dist_backend = "nccl"
    device_type = "cuda"
    device_control_env_var = "CUDA_VISIBLE_DEVICES"

    # -- worker / config selection ---------------------------------------------------
    @classmethod
    def get_omni_ar_worker_cls(cls) -> str:
        return AR_WORKER_QUALNAME

    @classmethod
    def get_omni_generation_worker_cls(cls) -> str:
        # the non-AR "generation" stages (code2wav, vocoder) are outside this backend's path:
        # leave them to the reference's own worker
        return "vllm_omni.worker.gpu_generation_worker.GPUGenerationWorker"

    @classmethod
    def get_default_stage_config_path(cls) -> str:
        return "vllm_omni/platforms/rocm/stage_configs"

    @classmethod
    def get_diffusion_attn_backend_cls(cls, selected_backend: Optional[str], head_size: int) -> str:
        # diffusion is out of scope; keep the reference's portable default
        return "vllm_omni.diffusion.attention.backends.sdpa.SDPABackend"

    @classmethod
    def get_attn_backend_cls(cls, *args, **kwargs) -> str:
        return "vllm_omni_amd.layers.MI355XAttentionBackend"

    @classmethod
    def supports_torch_inductor(cls) -> bool:
        return False  # HIP streams and graphs instead of a tracing compiler

    # -- device API ------------------------------------------------------------------
    @classmethod
    def get_torch_device(cls, local_rank: Optional[int] = None) -> torch.device:
        return torch.device("cuda") if local_rank is None else torch.device("cuda", local_rank)

    @classmethod
    def get_device_count(cls) -> int:
        return torch.cuda.device_count()

    @classmethod
    def get_device_version(cls) -> Optional[str]:
        return torch.version.hip.split("-")[0] if torch.version.hip else None

    @classmethod
    def get_device_arch(cls, device: int = 0) -> Optional[str]:
        if not torch.cuda.is_available():
            return None
        return torch.cuda.get_device_properties(device).gcnArchName.split(":")[0]

    @classmethod
    def synchronize(cls) -> None:
        torch.cuda.synchronize()

    @classmethod
    def get_free_memory(cls, device: Optional[torch.device] = None) -> int:
        free, _ = torch.cuda.mem_get_info(device)
        return free

    @classmethod
    def get_total_memory(cls, device: Optional[torch.device] = None) -> int:
        _, total = torch.cuda.mem_get_info(device)
        return total

    @classmethod
    def set_device_control_env_var(cls, devices) -> None:
        os.environ["HIP_VISIBLE_DEVICES"] = str(devices)
        os.environ["CUDA_VISIBLE_DEVICES"] = str(devices)

    @classmethod
    def unset_device_control_env_var(cls) -> None:
        os.environ.pop("HIP_VISIBLE_DEVICES", None)
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)

    @classmethod
    def create_autocast_context(cls, *, device_type: str, dtype: torch.dtype, enabled: bool = True):
        if not enabled:
            return nullcontext()
        try:
            return torch.autocast(device_type=device_type, dtype=dtype, enabled=True)
        except (RuntimeError, TypeError, ValueError):
            return nullcontext()

    @classmethod
    def supports_cpu_offload(cls) -> bool:
        return True

    @classmethod
    def supports_float64(cls) -> bool:
        return True

    @classmethod
    def check_if_supports_dtype(cls, dtype: torch.dtype) -> None:
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(f"MI355X decode backend computes in bf16/fp16; got {dtype}")

    @classmethod
    def get_profiler_cls(cls) -> str:
        return "vllm_omni.profiler.omni_torch_profiler.OmniTorchProfilerWrapper"


def mi355x_platform_plugin() -> Optional[str]:
    """Activate only on gfx950 with the HIP library present; otherwise stay out of the way (None)."""
    if os.environ.get("VLLM_OMNI_AMD_DISABLE") == "1":
        return None
    try:
        if not torch.cuda.is_available() or torch.version.hip is None:
            return None
        if MI355XOmniPlatform.get_device_arch(0) not in SUPPORTED_ARCHS:
            return None
        from . import _C  # noqa: F401  (raises if the .so is missing: fail loudly, do not half-activate)
    except ImportError:
        raise
    except Exception:  # noqa: BLE001
        return None
    return PLATFORM_QUALNAME
