for gb in 0 1 4 16; do
  s=$(date +%s.%N)
  python - <<PY
import torch, time, os
t0=time.time()
torch.cuda.init(); x=torch.empty(1,device='cuda')
n=int($gb*(1<<30))
if n:
    y=torch.empty(n,dtype=torch.uint8,device='cuda'); y.zero_()
torch.cuda.synchronize()
print("gb=$gb ready after %.3f s"%(time.time()-t0), flush=True)
open("/tmp/t_end","w").write(repr(time.time()))
os._exit(0)
PY
  e=$(date +%s.%N)
  python -c "
t=float(open('/tmp/t_end').read()); print('  exit took %.3f s, whole process %.3f s' % ($e - t, $e - $s))"
done
