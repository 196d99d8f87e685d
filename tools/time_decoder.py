import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
specs=syn.specs_for("nerf3"); sd=syn.full_state_dict("nerf3")
dec=HipSdfDecoder(sd,256,3,"nerf",device="cuda:0")
dec.set_sample(torch.from_numpy(syn.latent_code(0)))
import os
for N, var in ((64,"-"),(128,"-"),(256,"-")):
  os.environ["ASDF_K1_VARIANT"]=var
  for it in range(2):
        torch.cuda.synchronize(); t=time.time()
        h,o,b=dec.decode_grid(N,[-1,-1,-1],2.0/(N-1))
        torch.cuda.synchronize(); dt=time.time()-t
        fl=N**3*2*1573888
        print("variant %s N=%d pass %.4f s  alg %.1f TF/s exec %.1f TF/s (%.1f%% of 157.3) neg=%s"%(var,N,dt,fl/dt/1e12,N**3*2*1057792/dt/1e12,N**3*2*1057792/dt/1e12/1.573,(int(b[6]),int(b[14]))),flush=True)
