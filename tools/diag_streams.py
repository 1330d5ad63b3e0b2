import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda")
side = torch.cuda.Stream()
x = torch.randn(1 << 20, device=dev)
def step():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        y = x * 2
        if "event" in mode:
            ev = torch.cuda.Event(); ev.record(side)
    if "event" in mode:
        cur.wait_event(ev)
    z = x + 1
    cur.wait_stream(side)
    if "record" in mode:
        y.record_stream(cur)
    return y + z
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(out.sum()))
