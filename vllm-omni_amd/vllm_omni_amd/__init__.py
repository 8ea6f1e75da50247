"""MI355X-native (gfx950) decode hot path for vLLM-Omni's talker loop.

Host side is Python on PyTorch-ROCm; all arithmetic on the path runs in hand-written
HIP kernels behind the C ABI in ``include/vllm_omni_amd.h`` (``lib/libvllm_omni_amd.so``).
There is no CPU fallback: importing :mod:`vllm_omni_amd.ops` raises if the library is
missing.  See DESIGN.md for the path, its boundary and the kernels.
"""

__version__ = "0.1.0"


def register_platform():
    """Entry point for the ``vllm_omni.platform_plugins`` group.

    Mirrors the contract of vllm_omni/plugins/__init__.py:24-58 and
    vllm_omni/platforms/__init__.py:133-165: return the platform class qualname when
    this platform can activate, else ``None``.
    """
    from .platform import mi355x_platform_plugin

    return mi355x_platform_plugin()
