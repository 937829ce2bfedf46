import os,sys,torch
sys.path.insert(0,".")
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine, sample_params
mode=os.environ["GVC_TD_DTYPE"]; B=int(os.environ.get("GVC_TD_B","1"))
dims=gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w=synth.make_weights(1,synth.gpt_weight_spec(dims),device="cuda")
eng=GptEngine(dims,max_slots=8,max_rows=4096,weight_dtype=mode); eng.bind(w)
Tc=13; n=64
cond=synth.uniform(1,"c",(B,32,1024),1.0).cuda(); codes=synth.integers(1,"k",(B,Tc),256).cuda().int()
slots=torch.arange(B,device="cuda",dtype=torch.int32); prefix=eng.prefix_embeddings(cond,codes); P=prefix.shape[1]
sp=sample_params(dict(gcfg.DEFAULT_SAMPLING,top_k=1),1026,-1)
for _ in range(3):
    ids=torch.ones(B,P+1+n+8,device="cuda",dtype=torch.int32); ids[:,P]=1024
    il=torch.full((B,),P+1,device="cuda",dtype=torch.int32); fin=torch.zeros(B,device="cuda",dtype=torch.int32)
    toks=torch.zeros(B,n,device="cuda",dtype=torch.int32); lats=torch.zeros(B,n,1024,device="cuda")
    eng.prefill(slots,prefix,want_outputs=False)
    eng.generate(slots,ids,il,fin,sp,0,n,toks,lats); torch.cuda.synchronize()
