import os, sys
sys.path.insert(0, "/root/repo")
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine
for args, name in ((gcfg.TINY_MODEL_ARGS, "tiny"),):
    dims = gcfg.gpt_dims(args)
    w = synth.make_weights(5, synth.gpt_weight_spec(dims), device="cuda")
    for mode in ("fp32",):
        for B, Tc in ((8, 75), (6, 75)):
            outs = []
            for strip in ("0", "1"):
                os.environ["GVC_STRIP_PREFILL"] = strip
                eng = GptEngine(dims, max_slots=8, max_rows=2048, weight_dtype=mode)
                eng.bind(w)
                cond = synth.uniform(51, "c", (B, 32, dims["d_model"]), 1.0).cuda()
                codes = synth.integers(51, "k", (B, Tc), 256).cuda().int()
                prefix = eng.prefix_embeddings(cond, codes)
                lg, lat = eng.prefill(torch.arange(B, device="cuda", dtype=torch.int32), prefix)
                outs.append((lg.clone(), lat.clone()))
                eng.close()
            print(name, mode, B, Tc, "max |dlogits|", (outs[0][0] - outs[1][0]).abs().max().item(), "max |dlat|", (outs[0][1] - outs[1][1]).abs().max().item(), flush=True)
