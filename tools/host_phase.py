import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from trex_amd import capi, synth, weights
from trex_amd.pipeline import Pipeline, Lane
W,H,n_ind,_=synth.CONFIGS["C4"]; B=256
frames,bg=synth.batch_torch("C4",B,"cuda")
fc=torch.stack([frames,frames,frames,torch.full_like(frames,255)],dim=-1).contiguous()
host=[np.ascontiguousarray(fc[i].cpu().numpy()) for i in range(B)]
st=weights.synthetic_state(100,4242)
pipe=Pipeline(W,H,n_ind,B,100,bg,weights.pack_blob(st,100),bgra_in=True,host_frames=host,pipeline=(os.environ.get("ONE_LANE","0")!="1"))
T={"detect":[], "identify":[], "drain":[]}
od,oi,odr=Lane.detect,Lane.identify,Lane.drain
def wrap(name,f):
    def g(self,*a):
        t=time.perf_counter(); r=f(self,*a); T[name].append(time.perf_counter()-t); return r
    return g
Lane.detect=wrap("detect",od); Lane.identify=wrap("identify",oi); Lane.drain=wrap("drain",odr)
pipe.run(2,0)
for ln in pipe.lanes: ln.seg.profile_enable(True); ln.seg.profile_reset()
for k in T: T[k].clear()
torch.cuda.synchronize(); t0=time.perf_counter(); pipe.run(12,0); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("ms/step %.1f"%(dt/12*1e3), {k:[round(x*1e3,1) for x in v] for k,v in T.items()})
for k,ln in enumerate(pipe.lanes):
    print("lane",k,"copy (ms, frames)",ln.seg.profile_read(capi.STAGE_UPLOAD_COPY),"dma",ln.seg.profile_read(capi.STAGE_UPLOAD_DMA))
pipe.close()
