#!/usr/bin/env python3
"""How many times can a HIP interprocess event be recorded by its owner and waited for by a process that opened its handle?
Two processes on one GPU: A creates the event (scone_ipc_event_create), B opens it; N rounds of {A records on its stream,
host barrier, B makes its stream wait, B synchronises, host barrier}.  Prints the round at which either side first fails.
(The sdma transport of the sharded step signals "sent" / "reduced" with such events: the split-phase soak found
hipStreamWaitEvent failing with "invalid argument" after a few dozen steps.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def table():
    from scone_amd.hip_backend import SconeTable
    return SconeTable(max_n=3, dim=0, device="cuda:0") if False else None


def worker(rank, conn, bar, n, mode, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        torch.cuda.set_device(0)
        from scone_amd import EmbeddingCache, NGramExtractor
        from scone_amd import synthetic as S
        keys, lens = S.make_keys(60_000, S.GPT2_VOCAB, 3, seed=11)
        t = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys, lens, max_n=3), 64, table_format="int8").table
        x = torch.zeros(1 << 20, device="cuda")
        if rank == 0:
            ev, h = t.ipc_event_create()
            conn.send(h)
        else:
            ev = t.ipc_event_open(conn.recv())
        fail = None
        for i in range(n):
            try:
                if rank == 0:
                    x.add_(1.0)
                    t.ipc_event_record(ev)
                bar.wait()
                if rank == 1 and (mode == "every" or i % 2 == 0):        # "every": each record is waited for; else every other one
                    t.ipc_event_wait(ev)
                    torch.cuda.synchronize()
                bar.wait()
            except Exception as e:
                fail = (i, repr(e)[:200])
                break
        q.put((rank, fail))
        if fail is not None:
            bar.abort()
    except Exception as e:
        q.put((rank, ("setup", repr(e)[:300])))
        try:
            bar.abort()
        except Exception:
            pass


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mode = sys.argv[2] if len(sys.argv) > 2 else "every"
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    bar, q = ctx.Barrier(2), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, (a, b)[r], bar, n, mode, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    for _ in ps:
        try:
            res.append(q.get(timeout=120))
        except Exception as e:
            res.append(("timeout", repr(e)))
    for p in ps:
        p.join(timeout=20)
        if p.is_alive():
            p.terminate()
    print(json.dumps({"rounds": n, "mode": mode, "results": res}))


if __name__ == "__main__":
    main()
