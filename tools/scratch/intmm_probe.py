import torch, time
dev='cuda:0'
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for m,n,k in ((512,4608,3584),(512,3584,3584),(512,37888,3584),(512,3584,18944),(256,37888,3584),(512,2048,1024)):
    a=torch.randint(-127,127,(m,k),dtype=torch.int8,device=dev)
    w=torch.randint(-127,127,(n,k),dtype=torch.int8,device=dev)
    wt=w.t()   # [k,n] column-major view
    try:
        out=torch._int_mm(a,wt)
        ref=(a[:8].float()@w.float().t())
        ok=torch.equal(out[:8].float(),ref)
        t=timeit(lambda: torch._int_mm(a,wt))
        wc=w.t().contiguous()
        t2=timeit(lambda: torch._int_mm(a,wc))
        print(m,n,k,'int_mm ok',ok,f'{t:.1f} us (w.t view) {t2:.1f} us (contig [k,n])', f'{n*k/t/1e6:.2f} TB/s')
    except Exception as e:
        print(m,n,k,'int_mm failed',repr(e)[:200])
