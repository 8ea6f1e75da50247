"""CPU: the branches of scheduler.py / platform.py that only run where the reference stack (vllm + vllm_omni) is installed,
exercised against STAND-INS of the few upstream classes they bind to, injected into ``sys.modules`` and the modules
reloaded (VERDICT r3 "missing" #2 / ADVICE r3).

The stand-ins restate how upstream drives its KV cache manager, as far as the reference's files show it (they are test
scaffolding written here, not copies: vllm itself is not in this container):
  * ``vllm_omni.core.sched.omni_ar_scheduler.OmniARScheduler.__init__(*args, **kwargs)`` -> ``super().__init__`` of
    upstream's ``Scheduler(vllm_config, kv_cache_config, structured_output_manager, block_size, ...)``, which builds
    ``self.kv_cache_manager = KVCacheManager(kv_cache_config=..., max_model_len=..., hash_block_size=..., enable_caching=...,
    use_eagle=..., log_stats=..., enable_kv_cache_events=..., dcp_world_size=..., pcp_world_size=..., metrics_collector=...)``
    (omni_ar_scheduler.py:41-77; the keyword list is the one ``KVCacheManager._init_upstream`` mirrors);
  * ``schedule()`` -> ``get_computed_blocks(request)`` / ``allocate_slots(request, n, num_new_computed_tokens,
    new_computed_blocks, num_lookahead_tokens=, num_external_computed_tokens=, delay_cache_blocks=)`` -> ``None`` means
    preempt (omni_ar_scheduler.py:183-193 -> super().schedule()); ``get_block_ids`` (:628), ``free`` (:528),
    ``take_events`` (:480); stats: ``make_prefix_cache_stats()`` read as ``.requests / .queries / .hits / .reset``;
  * ``vllm_omni.platforms.interface.OmniPlatform(vllm.platforms.Platform)`` with ``_omni_enum`` and the classmethods of
    platforms/interface.py:26-165.
"""
import importlib
import sys
import types
from types import SimpleNamespace as NS

import pytest

from vllm_omni_amd.sched_output import Request, RequestStatus, SamplingParams

NB, BS = 10, 4


class _UpstreamKVCacheManager:
    """What upstream's scheduler builds first (then replaced): records its keyword arguments."""

    def __init__(self, **kw):
        self.kw = kw
        self.enable_caching = kw["enable_caching"]
        self.enable_kv_cache_events = kw["enable_kv_cache_events"]


class _UpstreamScheduler:
    """The slice of ``vllm.v1.core.sched.scheduler.Scheduler`` that touches the KV cache manager."""

    def __init__(self, vllm_config, kv_cache_config, structured_output_manager, block_size, mm_registry=None,
                 include_finished_set=False, log_stats=False):
        self.vllm_config = vllm_config
        self.cache_config = vllm_config.cache_config
        self.kv_cache_config = kv_cache_config
        self.block_size = block_size
        self.max_model_len = vllm_config.model_config.max_model_len
        self.log_stats = log_stats
        self.use_eagle = False
        self.dcp_world_size = self.pcp_world_size = 1
        self.kv_cache_manager = _UpstreamKVCacheManager(
            kv_cache_config=kv_cache_config, max_model_len=self.max_model_len, hash_block_size=block_size,
            enable_caching=bool(self.cache_config.enable_prefix_caching), use_eagle=False, log_stats=log_stats,
            enable_kv_cache_events=False, dcp_world_size=1, pcp_world_size=1, metrics_collector=None)
        self.waiting, self.running, self.requests = [], [], {}

    def add_request(self, request):
        self.requests[request.request_id] = request
        self.waiting.append(request)

    def schedule(self):
        """Upstream's call pattern on the manager (running requests first, then waiting ones with a prefix lookup)."""
        scheduled = {}
        for req in list(self.running):
            new = self.kv_cache_manager.allocate_slots(req, 1, num_lookahead_tokens=0)
            if new is None:                       # pool exhausted: preempt (free + back to waiting)
                self.running.remove(req)
                req.status = RequestStatus.PREEMPTED
                self.kv_cache_manager.free(req)
                req.num_computed_tokens = 0
                self.waiting.insert(0, req)
                continue
            scheduled[req.request_id] = (1, new.get_block_ids())
        while self.waiting:
            req = self.waiting[0]
            if req.num_computed_tokens == 0:
                hit, n_hit = self.kv_cache_manager.get_computed_blocks(req)
            else:
                hit, n_hit = self.kv_cache_manager.empty_kv_cache_blocks, 0
            n = req.num_tokens - req.num_computed_tokens - n_hit
            new = self.kv_cache_manager.allocate_slots(req, n, n_hit, hit, num_lookahead_tokens=0,
                                                       num_external_computed_tokens=0, delay_cache_blocks=False)
            if new is None:
                break
            self.waiting.pop(0)
            req.status = RequestStatus.RUNNING
            req.num_computed_tokens += n_hit
            self.running.append(req)
            scheduled[req.request_id] = (n, self.kv_cache_manager.get_block_ids(req.request_id))
        return scheduled

    def _free_blocks(self, request):              # upstream frees blocks only: hashes live on ITS Request
        self.kv_cache_manager.free(request)
        self.requests.pop(request.request_id, None)

    def make_stats(self):
        st = self.kv_cache_manager.make_prefix_cache_stats()
        return NS(prefix_cache_stats=st, hit_rate=(st.hits / st.queries if st.queries else 0.0), requests=st.requests,
                  reset=st.reset)


@pytest.fixture
def reference_stack(monkeypatch):
    """vllm / vllm_omni stand-ins in sys.modules; scheduler.py and platform.py reloaded on top of them."""
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m

    class Platform:                                # vllm.platforms.Platform: what the reference reads from it
        device_name = "rocm"
        device_type = "cuda"
        dispatch_key = "CUDA"
        dist_backend = "nccl"
        device_control_env_var = "CUDA_VISIBLE_DEVICES"

        @classmethod
        def check_if_supports_dtype(cls, dtype):
            return None

    class RocmPlatform(Platform):
        pass

    import enum

    class OmniPlatformEnum(enum.Enum):
        CUDA = "cuda"
        ROCM = "rocm"
        NPU = "npu"
        XPU = "xpu"
        MUSA = "musa"
        UNSPECIFIED = "unspecified"

    class OmniPlatform(Platform):                  # platforms/interface.py:26-165: abstract classmethods raise
        _omni_enum: OmniPlatformEnum

        def is_rocm(self):
            return self._omni_enum == OmniPlatformEnum.ROCM

        @classmethod
        def get_omni_ar_worker_cls(cls):
            raise NotImplementedError

        @classmethod
        def get_omni_generation_worker_cls(cls):
            raise NotImplementedError

        @classmethod
        def get_default_stage_config_path(cls):
            raise NotImplementedError

    class OmniARScheduler(_UpstreamScheduler):     # omni_ar_scheduler.py:41-77
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self.requests_needing_kv_transfer = {}
            self.waiting_for_transfer_free = set()
            self.active_kv_transfers = set()

    mod("vllm")
    mod("vllm.platforms", Platform=Platform)
    mod("vllm.platforms.rocm", RocmPlatform=RocmPlatform)
    mod("vllm_omni")
    mod("vllm_omni.core")
    mod("vllm_omni.core.sched")
    mod("vllm_omni.core.sched.omni_ar_scheduler", OmniARScheduler=OmniARScheduler)
    mod("vllm_omni.platforms")
    mod("vllm_omni.platforms.interface", OmniPlatform=OmniPlatform, OmniPlatformEnum=OmniPlatformEnum)
    for name, m in mods.items():
        monkeypatch.setitem(sys.modules, name, m)
    import vllm_omni_amd.platform as platform_mod
    import vllm_omni_amd.scheduler as scheduler_mod

    importlib.reload(scheduler_mod)
    importlib.reload(platform_mod)
    assert scheduler_mod.HAVE_REFERENCE_STACK and platform_mod.HAVE_REFERENCE_STACK
    yield NS(scheduler=scheduler_mod, platform=platform_mod, OmniARScheduler=OmniARScheduler, OmniPlatform=OmniPlatform,
             RocmPlatform=RocmPlatform)
    for name in mods:
        monkeypatch.delitem(sys.modules, name, raising=False)
    importlib.reload(scheduler_mod)
    importlib.reload(platform_mod)
    assert not scheduler_mod.HAVE_REFERENCE_STACK and not platform_mod.HAVE_REFERENCE_STACK


def _configs(caching=True):
    vc = NS(scheduler_config=NS(max_num_seqs=4, max_num_batched_tokens=64, enable_chunked_prefill=True),
            cache_config=NS(block_size=BS, enable_prefix_caching=caching),
            model_config=NS(max_model_len=64, omni_kv_config=None, async_chunk=False))
    kvc = NS(num_blocks=NB, kv_cache_groups=[NS(kv_cache_spec=NS(block_size=BS))])
    return vc, kvc


def test_scheduler_subclasses_the_reference_and_swaps_in_this_packages_manager(reference_stack):
    from vllm_omni_amd.kv_cache import KVCacheManager

    S = reference_stack.scheduler.MI355XARScheduler
    assert issubclass(S, reference_stack.OmniARScheduler)
    vc, kvc = _configs()
    s = S(vc, kvc, None, BS, log_stats=True)                       # upstream's positional order
    kv = s.kv_cache_manager
    assert type(kv) is KVCacheManager                               # built through _init_upstream ...
    assert (kv.num_blocks, kv.block_size, kv.max_model_len, kv.enable_caching, kv.log_stats) == (NB, BS, 64, True, True)
    assert s.waiting_for_transfer_free == set()                     # ... after the reference's own constructor ran
    s2 = S(vllm_config=vc, kv_cache_config=kvc, structured_output_manager=None, block_size=BS)   # keyword form
    assert s2.kv_cache_manager.num_blocks == NB and not s2.kv_cache_manager.log_stats


def test_upstream_schedule_calls_land_on_the_cxx_pool(reference_stack):
    """The inherited schedule() drives THIS package's manager with upstream's argument pattern: prefix lookup for a new
    request, allocate_slots(request, n, n_hit, hit_blocks, num_lookahead_tokens=, delay_cache_blocks=), None on
    exhaustion -> preemption, get_block_ids, free, stats with attribute access."""
    s = reference_stack.scheduler.MI355XARScheduler(*_configs(), None, BS, log_stats=True)
    kv = s.kv_cache_manager
    a = Request("a", list(range(9)), SamplingParams())
    s.add_request(a)
    out = s.schedule()
    assert out["a"] == (9, ([1, 2, 3],)) and kv.block_pool.get_num_free_blocks() == NB - 1 - 3
    a.num_computed_tokens = 9
    a.output_token_ids.append(5)
    b = Request("b", list(range(9)), SamplingParams())              # same prompt: hits a's two full blocks
    s.add_request(b)
    out = s.schedule()
    assert out["a"][0] == 1 and out["b"][0] == 1 and out["b"][1][0][:2] == [1, 2] and b.num_computed_tokens == 8
    st = s.make_stats()
    assert st.prefix_cache_stats.requests == 2 and st.prefix_cache_stats.hits == 8 and st.requests == 2 and not st.reset
    assert 0.0 < st.hit_rate < 1.0
    # exhaustion: None -> upstream preempts (free keeps the hash chain of a PREEMPTED request)
    big = Request("big", list(range(100, 123)), SamplingParams())
    s.add_request(big)
    s.schedule()
    assert "big" not in kv.req_to_blocks and big in s.waiting
    # finish: upstream's _free_blocks only calls free(); the manager itself drops the finished request's hash chain
    a.status = RequestStatus.FINISHED_STOPPED
    s._free_blocks(a)
    assert "a" not in kv._req_hashes and "a" not in kv.req_to_blocks
    assert kv.get_block_ids("b")[0][:2] == [1, 2]                   # still held by b
    assert kv.take_events() == [] and kv.reset_prefix_cache() is False   # blocks in use: upstream's contract
    assert kv.empty_kv_cache_blocks.get_block_ids() == ([],)


def test_platform_subclasses_the_reference_bases_and_names_this_packages_classes(reference_stack):
    P = reference_stack.platform.MI355XOmniPlatform
    assert issubclass(P, reference_stack.OmniPlatform) and issubclass(P, reference_stack.RocmPlatform)
    assert P().is_rocm() and P.dist_backend == "nccl" and P.device_type == "cuda"
    mod, cls = P.get_omni_ar_worker_cls().rsplit(".", 1)
    assert getattr(importlib.import_module(mod), cls).__name__ == "GPUARWorker"
    # every abstract classmethod of the interface is overridden (none falls through to NotImplementedError)
    for name in ("get_omni_ar_worker_cls", "get_omni_generation_worker_cls", "get_default_stage_config_path"):
        assert isinstance(getattr(P, name)(), str), name
    assert reference_stack.platform.PLATFORM_QUALNAME.endswith("MI355XOmniPlatform")
    from vllm_omni_amd.scheduler import SCHEDULER_QUALNAME

    m, c = SCHEDULER_QUALNAME.rsplit(".", 1)
    assert getattr(importlib.import_module(m), c) is reference_stack.scheduler.MI355XARScheduler
