"""Test infrastructure: screens INPUT seeds for the oracle-driven greedy tests of the one-launch steps on the CPU.  For every test
configuration it runs the oracle's generation loop on candidate seeds and prints the smallest top-1 / top-2 gap of the penalised scores
over all streams and steps; the tests then use a seed whose smallest gap is comfortable (>= 2e-3, the screen oracle/make_golden.py applies
to the reference fixtures) and assert token EQUALITY instead of tolerating flips at near-ties.

    python tests/screen_rows_seeds.py [first_seed] [n_seeds]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from genvc_amd import config as gcfg          # noqa: E402
from genvc_amd import synth                   # noqa: E402
from oracle import genvc_oracle as O          # noqa: E402

GREEDY = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
WIDE2 = dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=2)


def round_bf16(w):
    """test_gpu_gpt._round_bf16: the five streamed matrices of every block and mel_head rounded to bf16"""
    out = {}
    for k, v in w.items():
        if k.endswith(("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")) or k == "mel_head.weight":
            out[k] = v.to(torch.bfloat16).to(torch.float32)
        else:
            out[k] = v
    return out


def min_margin(w, dims, cond, codes, n):
    B, Tc = codes.shape
    ref_t, _, ref_logits = O.generate(w, dims, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    m = 1e9
    for i in range(n):
        ids = torch.cat([torch.ones(B, 32 + Tc + 2, dtype=torch.long), torch.full((B, 1), 1024), ref_t[:, :i]], 1)
        p = O.process_logits(ref_logits[i], ids, 2.0, 1.0, 0, 1.0)
        t2 = p.topk(2, -1)[0]
        m = min(m, float((t2[:, 0] - t2[:, 1]).min()))
    return m


# (label, model args, weight seed, storage mode, name of the cond / codes streams, B, Tc, n)
CONFIGS = [
    ("long_ctx 8rows_2chunks", WIDE2, 31, "fp32", ("cond", "codes"), 6, 120, 24),
    ("long_ctx 16rows_2chunks", WIDE2, 31, "fp32", ("cond", "codes"), 12, 150, 20),
    ("long_ctx 8rows_4chunks", WIDE2, 31, "fp32", ("cond", "codes"), 5, 300, 16),
    ("long_ctx 16rows_4chunks", WIDE2, 31, "fp32", ("cond", "codes"), 16, 300, 12),
    ("long_ctx 8rows_2chunks_16heads", dict(WIDE2, gpt_n_heads=16), 31, "fp32", ("cond", "codes"), 6, 120, 16),
    ("long_ctx 16rows_4chunks_16heads", dict(WIDE2, gpt_n_heads=16), 31, "fp32", ("cond", "codes"), 12, 300, 10),
    ("long_ctx 8rows_4chunks_8heads", dict(WIDE2, gpt_n_heads=8), 31, "fp32", ("cond", "codes"), 7, 300, 10),
    ("rows bf16 8_streams", WIDE2, 5, "bf16", ("cond_latents", "content_codes"), 8, 13, 24),
    ("rows bf16_kv 8_streams", WIDE2, 5, "bf16_kv", ("cond_latents", "content_codes"), 8, 13, 24),
    ("rows bf16_kv 16_rows_2_chunks", WIDE2, 5, "bf16_kv", ("cond_latents", "content_codes"), 12, 150, 20),
    ("rows bf16 8_rows_4_chunks", WIDE2, 5, "bf16", ("cond_latents", "content_codes"), 3, 300, 12),
    ("one stream bf16 fused", WIDE2, 5, "bf16", ("cond_latents", "content_codes"), 1, 13, 30),
    ("one stream bf16_kv fused", WIDE2, 5, "bf16_kv", ("cond_latents", "content_codes"), 1, 13, 30),
    ("one stream bf16_kv key_chunks", WIDE2, 5, "bf16_kv", ("cond_latents", "content_codes"), 1, 150, 24),
    ("one stream d512_hd128 bf16_kv", dict(WIDE2, gpt_n_model_channels=512, gpt_n_heads=4), 5, "bf16_kv", ("cond_latents", "content_codes"), 1, 120, 24),
    ("one stream d512_hd256 bf16", dict(WIDE2, gpt_n_model_channels=512, gpt_n_heads=2), 5, "bf16", ("cond_latents", "content_codes"), 1, 13, 40),
    ("gpt bf16 tiny B8", gcfg.TINY_MODEL_ARGS, 5, "bf16", ("cond_latents", "content_codes"), 8, 75, 40),
    ("gpt bf16 full B1", gcfg.DEFAULT_MODEL_ARGS, 5, "bf16", ("cond_latents", "content_codes"), 1, 13, 24),
    ("gpt bf16_kv tiny B8", gcfg.TINY_MODEL_ARGS, 5, "bf16_kv", ("cond_latents", "content_codes"), 8, 75, 40),
    ("gpt bf16_kv tiny B3", gcfg.TINY_MODEL_ARGS, 5, "bf16_kv", ("cond_latents", "content_codes"), 3, 75, 70),
    ("gpt bf16_kv full B1", gcfg.DEFAULT_MODEL_ARGS, 5, "bf16_kv", ("cond_latents", "content_codes"), 1, 13, 24),
    ("gpt bf16_kv full B8", gcfg.DEFAULT_MODEL_ARGS, 5, "bf16_kv", ("cond_latents", "content_codes"), 8, 13, 12),
]

if __name__ == "__main__":
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.set_num_threads(8)
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    for label, margs, wseed, mode, names, B, Tc, n in CONFIGS:
        if only not in label:
            continue
        dims = gcfg.gpt_dims(margs)
        w = synth.make_weights(wseed, synth.gpt_weight_spec(dims), device="cpu")
        if mode != "fp32":
            w = round_bf16(w)
            dims = dict(dims, kv_bf16=mode == "bf16_kv")
        best = (0.0, None)
        for seed in range(s0, s0 + ns):
            cond = synth.uniform(seed, names[0], (B, 32, dims["d_model"]), 1.0)
            codes = synth.integers(seed, names[1], (B, Tc), 256)
            m = min_margin(w, dims, cond, codes, n)
            if m > best[0]:
                best = (m, seed)
            if m >= 3e-3:
                break
        print(f"{label:36s} best input seed {best[1]}  min margin {best[0]:.2e}", flush=True)
