"""Times ll_sample_token_topk_bf16 on one row of synthetic logits: with the dbg tap (counts the whole row) and without
(top-k lower bound), and with the workspace (candidates launch + finish launch).  python tools/sample_time.py [V] [top_k] [B]"""
import os
import sys

os.environ.setdefault("LLAMOLE_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamole_amd import _lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 152064
top_k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load()
d = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
logits = (torch.randn(B, V, generator=g) * 2.5).to(torch.bfloat16).to(d)
seed = torch.tensor([1234], dtype=torch.long, device=d)
eos = torch.full((32,), -1, dtype=torch.long, device=d)
done = torch.zeros(B, dtype=torch.uint8, device=d)
tok = torch.zeros(B, dtype=torch.long, device=d)
out = torch.zeros(B, 4096, dtype=torch.long, device=d)
step = torch.zeros(B, dtype=torch.long, device=d)
dbg = torch.zeros(B, 4, dtype=torch.int64, device=d)
st = torch.cuda.current_stream().cuda_stream


ws = torch.zeros(int(lib.ll_sample_workspace_bytes(B)), dtype=torch.uint8, device=d)


def run(with_dbg, k=top_k, greedy=0):
    split = with_dbg == "split"
    rc = lib.ll_sample_token_topk_ws_bf16(logits.data_ptr(), logits.stride(0), B, V, 1.0 / 0.7, 0.9, k, greedy, seed.data_ptr(), eos.data_ptr(), 32, 0,
                                          done.data_ptr(), tok.data_ptr(), out.data_ptr(), out.stride(0), out.shape[1], step.data_ptr(), None, None,
                                          0, dbg.data_ptr() if with_dbg is True else None, ws.data_ptr() if split else None, ws.numel(), st)
    assert rc == 0


def timed(with_dbg, k=top_k, greedy=0, iters=300):
    step.zero_()
    for _ in range(20):
        run(with_dbg, k, greedy)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step.zero_()
    e0.record()
    for _ in range(iters):
        run(with_dbg, k, greedy)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out[:, :iters].clone()


for name, wd, k, gr in (("whole row (dbg tap)", True, top_k, 0), ("top-k bound", False, top_k, 0), ("top-k split (2 launches)", "split", top_k, 0),
                        ("top_k off", False, 0, 0), ("greedy", False, 0, 1)):
    us, toks = timed(wd, k, gr)
    line = f"{name:26s} {us:7.2f} us per launch (back to back, B={B}, V={V})"
    print(line)
    if name.startswith("whole"):
        ref = toks
    elif name.startswith("top-k"):
        print("   tokens equal to the whole-row path:", bool((toks == ref).all()), "distinct tokens:", int(toks.unique().numel()))
