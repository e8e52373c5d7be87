# dev soak: the fuzz tests of tests/test_gpu_fuzz.py with seeds beyond the committed range (GPU box): python tools/soak_fuzz.py
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest, importlib
import tests.test_gpu_fuzz as f
class MP:
    def __init__(self): self.saved=[]
    def setenv(self,k,v): self.saved.append((k,os.environ.get(k))); os.environ[k]=v
    def delenv(self,k,raising=True): self.saved.append((k,os.environ.get(k))); os.environ.pop(k,None)
    def setattr(self,*a,**k): raise RuntimeError
    def undo(self):
        for k,v in reversed(self.saved):
            if v is None: os.environ.pop(k,None)
            else: os.environ[k]=v
bad=0
for seed in range(40, 240):
    for method in ("auto","hash"):
        mp=MP()
        try: f.test_kernel_map_fuzz.__wrapped__(seed, method, mp) if hasattr(f.test_kernel_map_fuzz,'__wrapped__') else f.test_kernel_map_fuzz(seed, method, mp)
        except Exception as e: bad+=1; print("KMAP FAIL", seed, method, repr(e)[:200])
        finally: mp.undo()
for seed in range(32, 160):
    try: f.test_conv_fwd_bwd_fuzz(seed)
    except Exception as e: bad+=1; print("CONV FAIL", seed, repr(e)[:200])
print("soak done, failures:", bad)
