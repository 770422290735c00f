"""Steady-state cost of one greedy decode step (L6 H8 d768, 32 sequences, ~100 cached positions each), kernel by
kernel and per launch-profiler class.  R4D_LIB_PATH=tools/_bin/librag4dyg_dbg<n>.so times an ablated build
(tools/kc_ablate.sh gemm_skinny.hip SK_DBG n).    python tools/decode_step_bench.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import gpt2_ref                                   # noqa: E402  (weights only)
from rag4dyg_amd import _lib                                  # noqa: E402
from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModel      # noqa: E402

dev = torch.device("cuda:0")
L, H, d, V, B, T = 6, 8, 768, 1800, 32, 100
sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=4, random_affine=True)
m = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
m.load_state_dict(sd, strict=False); m.tie_weights()
tr = m.to(dev).eval().transformer
dec = tr.greedy_decoder(B, 384)
dec.use_graph = False
dec.cache.normal_()
dec.last.normal_()
dec.lens.fill_(T); dec.active.fill_(1); dec.gen_len.zero_()
dec.params.copy_(torch.tensor([10 ** 6, 10 ** 6, 0, 0, 0, 0, 0, 0], dtype=torch.int32))
lib = _lib.load()
dec._steps(5); torch.cuda.synchronize()
dec.lens.fill_(T); dec.gen_len.zero_()
t0 = time.perf_counter(); dec._steps(100); torch.cuda.synchronize()
print(os.environ.get("R4D_LIB_PATH", "product build"), "step wall us", round((time.perf_counter() - t0) / 100 * 1e6, 1))
if "-v" in sys.argv:
    dec.lens.fill_(T); dec.gen_len.zero_()
    lib.r4d_profile_enable(1)
    dec._steps(20); torch.cuda.synchronize()
    for c in range(lib.r4d_profile_num_classes()):
        ms, n, wk = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        lib.r4d_profile_read(c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(wk))
        if n.value:
            print("   ", lib.r4d_profile_class_name(c).decode(), n.value / 20, "launches/step", round(1e3 * ms.value / n.value, 2), "us each (HIP events)")
