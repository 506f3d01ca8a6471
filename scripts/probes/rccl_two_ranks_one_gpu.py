"""Probe: can libsumk's own RCCL entry (sumk_comm_init / sumk_allreduce_flat, SUMK_RCCL_DIRECT=1) run TWO ranks on ONE GPU?
NCCL >= 2.5 / RCCL refuse two ranks on the same device; this prints what actually happens (each rank under a timeout)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp


def run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SUMK_RCCL_DIRECT="1")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarizer_amd.training import RcclDirect
        torch.cuda.set_device(0)
        try:
            r = RcclDirect.get()
            buf = torch.full((1024,), float(rank + 1), device="cuda:0")
            r.all_reduce(buf)
            torch.cuda.synchronize()
            q.put((rank, "ok", float(buf[0].item())))
        except Exception as e:      # noqa: BLE001
            q.put((rank, "error", f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=run, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    out = []
    try:
        for _ in ps: out.append(q.get(timeout=120))
    except Exception as e:      # noqa: BLE001
        out.append(("?", "timeout", repr(e)))
    for p in ps:
        p.join(timeout=10)
        if p.is_alive(): p.kill()
    for o in sorted(out, key=str): print(o)
