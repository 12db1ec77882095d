import sys, time
sys.path.insert(0, "/root/repo")
import torch, stochopy_amd as sa
b=[[-5.12,5.12]]*128
def run(m):
    torch.cuda.synchronize(); t=time.perf_counter()
    r=sa.optimize.minimize(sa.factory.rosenbrock,b,method="de",options={"maxiter":m,"popsize":4096,"seed":0,"updating":"deferred","ftol":-1.0,"xtol":0.0})
    torch.cuda.synchronize(); return time.perf_counter()-t, r
run(3)
t1,r1=run(5); t2,r2=run(25)
print("legacy-stream DE M: %.1f ms/generation (%.3e evals/s), fun %r" % ((t2-t1)/(r2.nit-r1.nit)*1e3, 4096/((t2-t1)/(r2.nit-r1.nit)), r2.fun))
