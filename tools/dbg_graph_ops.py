import sys, torch, time
dev = torch.device("cuda:0")
B = 4096
scores = torch.randn(B, B, device=dev)
def run(name, fn, n=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for i in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(name, "ok", float(out.float().sum()), flush=True)
which = sys.argv[1]
diag = torch.eye(B, dtype=torch.bool, device=dev)
if which == "rand":
    run("rand", lambda: torch.rand((B, B), device=dev).masked_fill(diag, -1.0))
if which == "topk":
    run("topk", lambda: torch.topk(scores, k=20, dim=1).indices)
if which == "randtopk":
    run("randtopk", lambda: torch.topk(torch.rand((B, B), device=dev).masked_fill(diag, -1.0), k=20, dim=1).indices)
if which == "eye":
    run("eye", lambda: torch.eye(B, dtype=torch.bool, device=dev))
if which == "ce":
    lg = torch.randn(B, 21, device=dev, requires_grad=True)
    def f():
        t = torch.zeros(B, dtype=torch.long, device=dev)
        l = torch.nn.functional.cross_entropy(lg, t); l.backward(); return l
    run("ce", f)
if which in ("inbatch", "inbatch_hard"):
    sys.path.insert(0, ".")
    from torch_rechub_amd.utils.match import gather_inbatch_logits, inbatch_negative_sampling
    U = torch.randn(B, 64, device=dev, requires_grad=True)
    I = torch.randn(B, 64, device=dev, requires_grad=True)
    def f():
        u = torch.nn.functional.normalize(U, p=2, dim=1); it = torch.nn.functional.normalize(I, p=2, dim=1)
        scores = torch.matmul(u, it.t())
        neg = inbatch_negative_sampling(scores, neg_ratio=20, hard_negative=(which == "inbatch_hard"))
        logits = gather_inbatch_logits(scores, neg)
        loss = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device=dev))
        U.grad = None; I.grad = None
        loss.backward()
        return loss.detach()
    run(which, f)
