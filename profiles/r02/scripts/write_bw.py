import torch
dev=torch.device("cuda")
def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for mb in (43, 171, 686, 2744):
    n = mb*1000*1000//4
    C = torch.empty(n, device=dev); X = torch.rand(n, device=dev)
    t_fill = timeit(lambda: C.fill_(1.0))
    t_copy = timeit(lambda: C.copy_(X))
    t_read = timeit(lambda: X.sum())
    print("%5d MB: fill %.1f us = %.2f TB/s written | copy %.1f us = %.2f TB/s (read+write) | sum-reduce read %.1f us = %.2f TB/s" %
          (mb, t_fill, mb/t_fill, t_copy, 2*mb/t_copy, t_read, mb/t_read))
