import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, bench
torch.cuda.set_device(0)
L = bench.build_learner(1024, 32, 'cuda')
real = torch.rand(32, 3, 1024, 1024, device='cuda') * 2 - 1
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_alloc0 = torch.cuda.memory_stats()['num_device_alloc']
    bench.one_step(L, real)
    torch.cuda.synchronize()
    print(f'step {i}: {1e3*(time.perf_counter()-t0):.1f} ms, device allocs +{torch.cuda.memory_stats()["num_device_alloc"]-n_alloc0}, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB', flush=True)
