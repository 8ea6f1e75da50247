import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from vllm_omni_amd.model_runner import GPUARModelRunner
from vllm_omni_amd.talker import TalkerConfig
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = TalkerConfig.from_json("qwen2_5_omni_talker")
runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=256, max_model_len=4112, max_num_batched_tokens=256, block_size=16, seed=0)
runner.load_model()
sched = bench.SyntheticScheduler(256, 4096, 16, cfg.vocab_size, 0)
runner.initialize_kv_cache(sched.num_blocks); bench.fill_kv_cache(runner, 1); runner._bench_sched = sched
pending = None
for i in range(10):
    t0 = time.perf_counter()
    so = sched.schedule(); t1 = time.perf_counter()
    runner.execute_model(so); t2 = time.perf_counter()
    out = runner.sample_tokens(async_output=True); t3 = time.perf_counter()
    if pending is not None: pending.get_output()
    t4 = time.perf_counter(); pending = out
    print(f"step {i}: schedule {1e3*(t1-t0):.2f} exec {1e3*(t2-t1):.2f} sample {1e3*(t3-t2):.2f} get_output(prev) {1e3*(t4-t3):.2f} ms")
pending.get_output()
