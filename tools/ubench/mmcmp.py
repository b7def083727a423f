import torch, time
M,N=20000,512
Z=torch.randn((N,M),dtype=torch.float64,device='cuda')
def t(fn,reps=50):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e6
us=t(lambda: torch.mm(Z,Z.T)); print(f"torch.mm Z'Z 20000x512 f64: {us:.1f} us {2*M*N*N/us/1e6:.1f} TF")
for (m,n,k) in [(4096,4096,4096),(8192,8192,8192),(512,512,200000),(4096,4096,65536)]:
    A=torch.randn((m,k),dtype=torch.float64,device='cuda'); B=torch.randn((n,k),dtype=torch.float64,device='cuda')
    us=t(lambda: torch.mm(A,B.T),10); print(f"torch.mm {m}x{n}x{k} f64 NT: {us:.1f} us {2*m*n*k/us/1e6:.1f} TF")
    Bt=B.T.contiguous()
    us=t(lambda: torch.mm(A,Bt),10); print(f"torch.mm {m}x{n}x{k} f64 NN: {us:.1f} us {2*m*n*k/us/1e6:.1f} TF")
    del A,B,Bt
