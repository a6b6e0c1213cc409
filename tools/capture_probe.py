#!/usr/bin/env python3
"""dev (one-off): which multi-stream capture patterns does hipStreamEndCapture (ROCm 7.2, torch 2.10) accept?  Each pattern runs in
its own child process (a host-side crash of one must not take the others down); trivial torch kernels unless a pattern says
otherwise.  Prints one line per pattern: OK / the exception / the child's return code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ["two_branches", "three_from_origin", "nested_fork", "sibling_events", "three_sst_kernels", "three_long_chains", "four_from_origin",
            "rccl_world1_async", "rccl_world1_two_colls"]
if os.environ.get("PROBE2"):
    PATTERNS = ["nested_join_origin", "nested_join_origin_then_work", "nested_fork_no_tail", "nested_event_join", "rccl_side_wait_on_origin",
                "rccl_two_groups"]


def child(name):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "srgan-st_amd")]
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = [torch.ones(1 << 16, device=dev) for _ in range(6)]
    streams = [torch.cuda.Stream() for _ in range(5)]
    if name.startswith("rccl"):
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        import torch.distributed as td
        td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        td.all_reduce(x[0])
        torch.cuda.synchronize()

    def work(t, n=3):
        for _ in range(n):
            t.mul_(1.0001)

    def body():
        main = torch.cuda.current_stream()
        if name == "two_branches":
            s = streams[0]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                work(x[1])
            work(x[0])
            main.wait_stream(s)
        elif name in ("three_from_origin", "four_from_origin", "three_long_chains"):
            k = 3 if name == "four_from_origin" else 2
            for i in range(k):
                streams[i].wait_stream(main)
                with torch.cuda.stream(streams[i]):
                    work(x[i + 1], 150 if name == "three_long_chains" else 3)
            work(x[0], 150 if name == "three_long_chains" else 3)
            for i in range(k):
                main.wait_stream(streams[i])
        elif name == "nested_fork":
            a, b = streams[0], streams[1]
            a.wait_stream(main)
            with torch.cuda.stream(a):
                work(x[1])
                b.wait_stream(a)
                with torch.cuda.stream(b):
                    work(x[2])
                work(x[1])
                a.wait_stream(b)
            work(x[0])
            main.wait_stream(a)
        elif name in ("nested_join_origin", "nested_join_origin_then_work"):
            a, b = streams[0], streams[1]
            a.wait_stream(main)
            with torch.cuda.stream(a):
                work(x[1])
                b.wait_stream(a)                  # second-level fork ...
                with torch.cuda.stream(b):
                    work(x[2])
                work(x[1])
            work(x[0])
            main.wait_stream(a)
            main.wait_stream(b)                   # ... joined into the ORIGIN stream
            if name.endswith("then_work"):
                work(x[0])
        elif name == "nested_fork_no_tail":
            a, b = streams[0], streams[1]
            a.wait_stream(main)
            with torch.cuda.stream(a):
                work(x[1])
                b.wait_stream(a)
                with torch.cuda.stream(b):
                    work(x[2])
                a.wait_stream(b)                  # joined into the intermediate stream, which does nothing afterwards
            work(x[0])
            main.wait_stream(a)
        elif name == "nested_event_join":
            a, b = streams[0], streams[1]
            a.wait_stream(main)
            with torch.cuda.stream(a):
                work(x[1])
                b.wait_stream(a)
                with torch.cuda.stream(b):
                    work(x[2])
                    e = torch.cuda.Event()
                    e.record(b)
                work(x[1])
                a.wait_event(e)                   # the intermediate stream depends on its child through an event, the child ALSO joins the origin
                work(x[1])
            work(x[0])
            main.wait_stream(a)
            main.wait_stream(b)
        elif name == "rccl_side_wait_on_origin":
            import torch.distributed as td
            s = streams[0]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                work(x[1])
                h = td.all_reduce(x[1], async_op=True)
                work(x[2])
            work(x[0])
            main.wait_stream(s)
            h.wait()                              # on the origin stream
            work(x[0])
        elif name == "rccl_two_groups":
            import torch.distributed as td
            pg2 = td.new_group([0])
            td.all_reduce(x[3], group=pg2)
            s = streams[0]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                work(x[1])
                h1 = td.all_reduce(x[1], async_op=True)
                work(x[2])
                h2 = td.all_reduce(x[2], async_op=True)
            work(x[0])
            h0 = td.all_reduce(x[0], async_op=True, group=pg2)
            h0.wait()
            work(x[0])
            main.wait_stream(s)
            h1.wait()
            h2.wait()
            work(x[0])
        elif name == "sibling_events":
            a, b = streams[0], streams[1]
            a.wait_stream(main)
            b.wait_stream(main)
            evs = []
            with torch.cuda.stream(a):
                for _ in range(4):
                    work(x[1], 1)
                    e = torch.cuda.Event()
                    e.record(a)
                    evs.append(e)
            with torch.cuda.stream(b):
                for e in evs:
                    b.wait_event(e)
                    work(x[2], 1)
            del evs
            work(x[0])
            main.wait_stream(a)
            main.wait_stream(b)
        elif name == "three_sst_kernels":
            from srganst import ops
            for i in range(2):
                streams[i].wait_stream(main)
                with torch.cuda.stream(streams[i]):
                    for _ in range(3):
                        x[i + 1] = ops.add(x[i + 1], x[i + 1])
            for _ in range(3):
                x[0] = ops.add(x[0], x[0])
            for i in range(2):
                main.wait_stream(streams[i])
        elif name == "rccl_world1_async":
            import torch.distributed as td
            s = streams[0]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                work(x[1])
                h = td.all_reduce(x[1], async_op=True)        # the process group's own stream: a third branch
                work(x[2])
                h.wait()
            work(x[0])
            main.wait_stream(s)
        elif name == "rccl_world1_two_colls":
            import torch.distributed as td
            s = streams[0]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                work(x[1])
                h1 = td.all_reduce(x[1], async_op=True)
                work(x[2])
                h2 = td.all_reduce(x[2], async_op=True)
            work(x[0])
            h0 = td.all_reduce(x[0], async_op=True)
            h0.wait()
            work(x[0])
            main.wait_stream(s)
            h1.wait()
            h2.wait()

    s0 = torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        body()                       # eager warm-up
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream(), capture_error_mode="thread_local"):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("OK", name, float(x[0][0]), flush=True)
    if name.startswith("rccl"):
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), p], capture_output=True, text=True, timeout=300)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith("OK")]
            tail = (r.stderr.strip().splitlines() or [""])[-1][:160]
            print(f"{p:24s} rc={r.returncode:4d} {'OK' if out else 'FAILED: ' + tail}", flush=True)
