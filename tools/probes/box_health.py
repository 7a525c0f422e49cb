import torch, time
a=torch.empty(1<<30, dtype=torch.uint8, device='cuda'); b=torch.empty_like(a)
for r in range(3):
    torch.cuda.synchronize(); t=time.time()
    for i in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt=time.time()-t
    print("copy GB/s (r+w)", 2*20*(1<<30)/dt/1e9)
x=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16); y=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)
for r in range(3):
    torch.cuda.synchronize(); t=time.time()
    for i in range(20): z=x@y
    torch.cuda.synchronize(); dt=time.time()-t
    print("bf16 gemm TFLOP/s", 20*2*8192**3/dt/1e12)
