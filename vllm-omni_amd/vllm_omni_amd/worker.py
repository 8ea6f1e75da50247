"""AR worker: one OS process per GPU / TP rank (SURVEY §8b "Worker class", §3.2).

Mirrors ``GPUARWorker`` (vllm_omni/worker/gpu_ar_worker.py:29-106) and ``OmniGPUWorkerBase``
(vllm_omni/worker/base.py:49-156): ``init_device`` (set device, create the RCCL groups BEFORE the
memory snapshot, refuse non-GPU devices), ``load_model``, ``determine_available_memory`` (KV budget),
``initialize_cache``, ``execute_model`` / ``sample_tokens`` pass-through, ``profile``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .distributed import init_distributed_environment
from .kv_cache import determine_available_memory, get_process_gpu_memory, num_blocks_for_budget
from .model_runner import GPUARModelRunner
from .talker import TalkerConfig


@dataclass
class WorkerConfig:
    model_config: TalkerConfig
    dtype: torch.dtype = torch.bfloat16
    tensor_parallel_size: int = 1
    world_size: int = 1
    gpu_memory_utilization: float = 0.9
    kv_cache_memory_bytes: Optional[int] = None
    block_size: int = 16
    max_num_seqs: int = 256
    max_model_len: int = 4096
    max_num_batched_tokens: int = 32768
    enforce_eager: bool = False
    engine_output_type: str = "latent"
    kv_layout: str = "NB2_HEAD_MAJOR"
    device_type: str = "cuda"
    distributed_backend: str = "nccl"
    seed: int = 0
    quantization: Optional[str] = None        # None | "int8" (W8A8 decoder linears, int8_config.py)
    dp_local_rank: int = 0                    # parallel_config.data_parallel_rank_local (gpu_ar_worker.py:40-60)
    # when the launcher already hands every worker its device index the DP offset must NOT be applied
    # (gpu_ar_worker.py:34-38: executor backend "ray" / "external_launcher", a ray DP backend, DP across nodes)
    distributed_executor_backend: Optional[str] = None
    data_parallel_backend: Optional[str] = None
    nnodes_within_dp: int = 1
    enable_prefix_caching: bool = False       # cache_config.enable_prefix_caching: builds OmniTensorPrefixCache (A14)
    max_reply_rows: int = 0                   # device pool of thinker_reply_part rows; 0 = max_num_seqs * 64
    code_predictor: object = None             # CodePredictorConfig for the Qwen3-Omni talker_mtp step


class GPUARWorker:
    """Constructed the way the reference constructs its workers: ``worker_cls(vllm_config=..., local_rank=...,
    rank=..., distributed_init_method=..., is_driver_worker=...)`` (upstream ``Worker.__init__``; the class is named
    by string, vllm_omni/engine/stage_init_utils.py:86-101).  ``vllm_config`` is any object with ``model_config``
    (``hf_config``, dtype, max_model_len, ...), ``cache_config``, ``parallel_config``, ``scheduler_config`` on it
    (mapped by ``compat.worker_config_from_vllm_config``); this package's own ``WorkerConfig`` is accepted too."""

    def __init__(self, vllm_config=None, local_rank: int = 0, rank: int = 0,
                 distributed_init_method: Optional[str] = None, is_driver_worker: bool = True, **kwargs):
        from .compat import is_vllm_config, worker_config_from_vllm_config

        if vllm_config is None:
            vllm_config = kwargs.pop("config", None)
        if isinstance(vllm_config, WorkerConfig):
            config = vllm_config
            self.vllm_config = None
        elif is_vllm_config(vllm_config):
            config = worker_config_from_vllm_config(vllm_config)
            self.vllm_config = vllm_config
        else:
            raise TypeError("GPUARWorker needs a VllmConfig-shaped object (model_config / cache_config / "
                            f"parallel_config / scheduler_config) or a WorkerConfig, got {type(vllm_config).__name__}")
        self.config = config
        self.local_rank = local_rank
        self.rank = rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.device: Optional[torch.device] = None
        self.model_runner: Optional[GPUARModelRunner] = None
        self.init_free_bytes = 0
        self.total_bytes = 0
        self._profiler = None

    def dp_adjusted_local_rank(self) -> int:
        """``DP_LOCAL_RANK * TP_PP_WORLD_SIZE + TP_LOCAL_RANK`` — applied in ``init_device`` and only under the
        reference's condition (gpu_ar_worker.py:34-49): not when ray / an external launcher (torchrun) already
        gave this worker its device index, not with a ray DP backend, not when DP spans nodes.  (Pipeline
        parallelism is refused by ``compat``, so TP_PP_WORLD_SIZE is the TP size.)"""
        c = self.config
        if (c.distributed_executor_backend in ("ray", "external_launcher") or c.data_parallel_backend == "ray"
                or c.nnodes_within_dp != 1):
            return self.local_rank
        return self.local_rank + c.dp_local_rank * c.tensor_parallel_size

    def init_device(self) -> None:
        if self.config.device_type != "cuda":
            # the reference refuses anything else the same way (gpu_ar_worker.py:89-90)
            raise RuntimeError(f"Not support device type: {self.config.device_type}")
        if not torch.cuda.is_available():
            raise RuntimeError("Not support device type: no ROCm GPU visible (the HIP kernels have no CPU path)")
        self.local_rank = self.dp_adjusted_local_rank()
        if self.local_rank >= torch.cuda.device_count():
            raise RuntimeError(f"DP adjusted local rank {self.local_rank} is out of bounds "
                               f"({torch.cuda.device_count()} visible devices)")
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        from .platform import MI355XOmniPlatform

        MI355XOmniPlatform.check_if_supports_dtype(self.config.dtype)
        # distributed groups first, so communicator buffers are inside the memory snapshot
        init_distributed_environment(self.config.world_size, self.rank, self.config.tensor_parallel_size,
                                     backend=self.config.distributed_backend,
                                     init_method=self.distributed_init_method, device=self.device)
        torch.cuda.empty_cache()
        self.init_free_bytes, self.total_bytes = torch.cuda.mem_get_info(self.device)
        self.model_runner = GPUARModelRunner(
            self.config.model_config, self.device, self.config.dtype, max_num_reqs=self.config.max_num_seqs,
            max_model_len=self.config.max_model_len, max_num_batched_tokens=self.config.max_num_batched_tokens,
            block_size=self.config.block_size, kv_layout=self.config.kv_layout,
            enforce_eager=self.config.enforce_eager, engine_output_type=self.config.engine_output_type,
            seed=self.config.seed, quantization=self.config.quantization,
            max_reply_rows=self.config.max_reply_rows or self.config.max_num_seqs * 64,
            code_predictor=self.config.code_predictor, enable_prefix_caching=self.config.enable_prefix_caching)

    def load_model(self, weights=None) -> None:
        """Build the model on the device; ``weights`` (an iterable of ``(checkpoint name, tensor)``, e.g. a
        safetensors iterator) is streamed through ``model.load_weights`` — without it the weights are random-init
        (this package ships no checkpoint loader: upstream's ``get_model_loader`` feeds the same iterable)."""
        self.model_runner.load_model()
        if weights is not None:
            self.model_runner.model.load_weights(weights)

    def get_model(self):
        return self.model_runner.get_model()

    @torch.inference_mode()
    def determine_available_memory(self) -> int:
        """Bytes available for the KV cache on this rank (base.py:77-156): requested =
        total * gpu_memory_utilization; available = requested - what this process already holds
        (weights + persistent buffers): the driver's per-process figure through amdsmi — the counterpart of
        the reference's NVML per-PID query — or, when that is unavailable, the allocator's own numbers (the
        profiling branch, base.py:134-154)."""
        torch.cuda.synchronize()
        free_now, total = torch.cuda.mem_get_info(self.device)
        # per-process accounting from the driver first (what the reference asks NVML for, base.py:107-133): correct
        # when other workers share the card; else what this process reserved / what left the free pool since init
        profiled = max(torch.cuda.memory_reserved(self.device), self.init_free_bytes - free_now)
        by_pid = get_process_gpu_memory(self.local_rank)
        self.memory_accounting = {"amdsmi_process_bytes": by_pid, "profiled_bytes": int(profiled)}
        # the allocator's reserved figure is a hard lower bound on what this process holds: a driver figure below it (a
        # wrong handle under an unusual device enumeration, an under-report) would oversize the KV cache
        reserved = int(torch.cuda.memory_reserved(self.device))
        used_by_process = max(by_pid, reserved) if by_pid is not None and by_pid > 0 else profiled
        self.memory_accounting["used_bytes"] = int(used_by_process)
        return determine_available_memory(total, self.config.gpu_memory_utilization, used_by_process,
                                          self.config.kv_cache_memory_bytes)

    def get_kv_cache_spec(self):
        return self.model_runner.get_kv_cache_spec()

    def determine_num_available_blocks(self) -> int:
        spec = self.get_kv_cache_spec()
        return num_blocks_for_budget(self.determine_available_memory(), self.config.model_config.num_hidden_layers, spec)

    def initialize_cache(self, num_gpu_blocks: int) -> None:
        self.model_runner.initialize_kv_cache(num_gpu_blocks)

    def compile_or_warm_up_model(self, batch_sizes: Optional[list[int]] = None) -> None:
        if not self.config.enforce_eager:
            self.model_runner.capture_model(batch_sizes)

    def execute_model(self, scheduler_output, intermediate_tensors=None):
        return self.model_runner.execute_model(scheduler_output, intermediate_tensors)

    def sample_tokens(self, grammar_output=None):
        return self.model_runner.sample_tokens(grammar_output)

    def profile(self, is_start: bool = True, profile_prefix: Optional[str] = None) -> None:
        """torch.profiler start/stop RPC (base.py:49-75); ranges show up as roctx under PyTorch-ROCm."""
        if is_start:
            self._profiler = torch.profiler.profile(
                activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA])
            self._profiler.__enter__()
        elif self._profiler is not None:
            self._profiler.__exit__(None, None, None)
            if profile_prefix:
                self._profiler.export_chrome_trace(f"{profile_prefix}_rank{self.rank}.json")
            self._profiler = None
