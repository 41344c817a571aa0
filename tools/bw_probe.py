import torch, time
dev='cuda'
n=308*1024*1024  # bf16 elements = 616 MB
a=torch.randn(n,device=dev).to(torch.bfloat16); b=torch.randn(n,device=dev).to(torch.bfloat16); c=torch.empty_like(a)
def t(f,name,bytes_):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print('%-10s %.3f ms  %.2f TB/s'%(name,ms,bytes_/ms/1e9))
t(lambda: c.copy_(a),'copy',2*n*2)
t(lambda: torch.add(a,b,out=c),'add',3*n*2)
t(lambda: torch.relu_(c),'relu_',2*n*2)
