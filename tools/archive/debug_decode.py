import torch, sys
sys.path.insert(0, ".")
from llamole_amd import e2e
from transformers import StaticCache
llm = e2e.build_llm("tiny", "cuda", torch.float32)
g = torch.Generator().manual_seed(0)
prompt = torch.randint(5, 1000, (2, 12), generator=g).cuda()
mask = torch.ones_like(prompt); mask[1, :4] = 0
B, P, NEW = 2, 12, 6
forced = torch.randint(5, 1000, (B, NEW), generator=g).cuda()

def run(mode):
    cache = StaticCache(config=llm.config, max_cache_len=P + NEW)
    full = torch.ones(B, P + NEW, dtype=torch.long, device="cuda"); full[:, :P] = mask
    pos_ids = (mask.cumsum(1) - 1).clamp_min(0)
    with torch.no_grad():
        llm(input_ids=prompt, attention_mask=full[:, :P], past_key_values=cache, cache_position=torch.arange(P, device="cuda"), position_ids=pos_ids, use_cache=True)
    tok = torch.zeros(B, 1, dtype=torch.long, device="cuda"); pos = torch.zeros(1, dtype=torch.long, device="cuda")
    posid = mask.sum(1, keepdim=True).clone()
    outs = []
    graph = None
    def step():
        with torch.no_grad():
            return llm(input_ids=tok, attention_mask=full, past_key_values=cache, cache_position=pos, position_ids=posid, use_cache=True).logits[:, -1]
    for t in range(NEW):
        tok.copy_(forced[:, t:t+1])
        if mode != "freeze_pos" or t == 0: pos.fill_(P + t)
        if t > 0 and mode != "freeze_posid": posid.add_(1)
        if mode == "graph":
            if graph is None:
                s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s): step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph): lg = step()
            graph.replay(); outs.append(lg.clone())
        else:
            outs.append(step().clone())
    return torch.stack(outs)
ref = run("eager")
print("eager ok", flush=True)
for mode in sys.argv[1:]:
    o = run(mode)
    d = (o - ref).abs().amax(dim=-1)
    print(mode, d.cpu().numpy().round(4).tolist())
