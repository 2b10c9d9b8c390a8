import torch,time
for mb in (67,268):
    x=torch.empty(mb*1024*1024//4,device='cuda')
    for _ in range(3): x.zero_()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): x.zero_()
    b.record(); torch.cuda.synchronize()
    us=a.elapsed_time(b)/20*1e3
    print(mb,'MB fill',round(us,1),'us',round(mb*1.048576/us*1e3/1e3,2),'TB/s')
