import time, torch
dev="cuda:0"
n,F=10_000_000,256
x=torch.randn(n,F,device=dev); g=torch.randn(n,F,device=dev)
def t(fn,reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(True),torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ref=torch.mm(x.t(),g)
print("mm(x.t(),g) TN  %.2f ms"%t(lambda: torch.mm(x.t(),g)))
print("mm(g.t(),x).t() %.2f ms"%t(lambda: torch.mm(g.t(),x).t()))
for b in (8,32,128,512,2048):
    m=n//b*b
    def f():
        p=torch.bmm(x[:m].view(b,m//b,F).transpose(1,2), g[:m].view(b,m//b,F)).sum(0)
        if m<n: p=p+torch.mm(x[m:].t(),g[m:])
        return p
    err=(f()-ref).abs().max().item()/ref.abs().max().item()
    print("bmm b=%4d       %.2f ms  relerr %.2e"%(b,t(f),err))
