"""What does an event cost the stream it is recorded on / waits on?  A chain of dependent small kernels on one stream, with
(a) nothing between them, (b) an event RECORD between them, (c) a record + another stream WAITING for it (and doing a tiny
kernel), (d) the main stream WAITING for an event of the other stream between them.  Microseconds per link of the chain."""
import sys, time, torch
dev = torch.device("cuda", 0)
x = torch.zeros(4096, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
ev = [torch.cuda.Event() for _ in range(64)]
y = torch.zeros(4096, device=dev)


def chain(n, mode):
    for i in range(n):
        x.add_(1.0)
        if mode == "record":
            ev[i % 64].record(main)
        elif mode == "record+sidewait":
            ev[i % 64].record(main)
            side.wait_event(ev[i % 64])
            with torch.cuda.stream(side):
                y.add_(1.0)
        elif mode == "mainwait":
            with torch.cuda.stream(side):
                y.add_(1.0)
                ev[i % 64].record(side)
            main.wait_event(ev[i % 64])


for mode in ("plain", "record", "record+sidewait", "mainwait", "plain"):
    chain(200, mode); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main); chain(2000, mode); e1.record(main); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 2000 * 1e3)
    print("%-16s %.2f us per link" % (mode, best))
