import numpy as np, torch, sys
sys.path.insert(0,'.')
from alignsdf_amd import synthetic as syn
from alignsdf_amd.marching_cubes import marching_cubes_device
from alignsdf_amd.networks.model import build_decoder
from alignsdf_amd.utils.mesh import decode_two_pass
from alignsdf_amd.utils.utils import hip_decoder_for
for name,tag,N,hb,ob in (("hand64","nerf3",64,True,False),("both9","both9",256,True,True)):
    g=np.load('tests/golden/ref_fullsize_%s.npz'%name)
    specs=syn.specs_for(tag)
    dec=build_decoder(specs,{k:torch.from_numpy(v) for k,v in syn.full_state_dict(tag).items()})
    lat=torch.from_numpy(syn.latent_code(0)).cuda()
    mano=obj=None
    if tag=="both9":
        m,o=syn.pose_inputs(0); mano={k:torch.from_numpy(v).cuda() for k,v in m.items()}; obj={k:torch.from_numpy(v).cuda() for k,v in o.items()}
    for math in ("f16x3","f32"):
        hip=hip_decoder_for(dec); hip.set_math(math)
        r=decode_two_pass(hb,ob,dec,lat,mano,obj,specs,N)
        for part,on in (("hand",hb),("obj",ob)):
            if not on: continue
            vol=r["vol_"+part]; v,f=marching_cubes_device(vol,0.0)
            sel=torch.from_numpy(g["probe_sel_%d"%N]).cuda()
            err=np.abs(vol.reshape(-1)[sel].cpu().numpy()-g["p2_%s_%d"%(part,N)]).max()
            print(name,math,part,"V/F",v.shape[0],f.shape[0],"ref",g["mc_%s_%d"%(part,N)].tolist(),"probe err %.2e"%err,"neg",int((vol<0).sum()),"ref neg",g["neg_count_%d"%N].tolist(),"near(1e-6) ours",int((vol.abs()<1e-6).sum()), "near 1e-5", int((vol.abs()<1e-5).sum()))
