"""Where do the 34.9 ms of the 2-rank rehearsal go (two half-shares of C4 time-slicing ONE GPU over gloo took 34.9 ms per
step against 21.7 ms for the whole model in one process)?  Per rank, with both ranks running at the same time: the assembly alone, the exchange
alone (gloo: device -> host -> gloo all_gather -> host -> device), both; and each rank's assembly with the other rank idle.  Also prints, from
the partition table, the work a rank does at 2 / 4 / 8 ranks (owned Gauss points, mortar vertices it evaluates incl. the cut interfaces' redundant
copies) -- the expected per-rank step time on separate GPUs.   Usage (one-GPU box): python tools/rehearsal_breakdown.py"""
import os, socket, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from goldfish_amd import _lib, geometry as G, sharding
    spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
    th, u = G.random_thickness(spec), G.smooth_displacement(spec, 0.5 * spec.h_th)
    part = sharding.partition_patches(spec, world)
    shard = sharding.shard_spec(spec, rank, world, part)
    A = sharding.shard_arrays(shard, th)
    D = _lib.DeviceModel(A, device=0)
    D.set_thickness(shard.to_local(np.concatenate(th))); D.set_u(shard.to_local(u, 3))

    class _Buf:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    R_loc = torch.as_tensor(_Buf(_lib.lib().gf_device_ptr(D.h, _lib.BUF_R), A.ndof), device="cuda")
    R_glob = torch.zeros(3 * shard.total_cp_global, dtype=torch.float64, device="cuda")

    def timed(fn, n=5):
        for _ in range(2): fn()
        D.sync(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        D.sync(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n * 1e3
        dist.barrier()
        return t
    asm = lambda: (D.assemble(_lib.ASM_ALL, sync=False), D.sync())
    exch = lambda: sharding.allgather_owned_rows(shard, R_loc, dist, 3, out=R_glob)
    out = {"rank": rank, "gauss_points": D.n_gauss_points, "mortar_points": D.n_mortar_points}
    out["assembly, both ranks busy"] = timed(asm)
    out["exchange alone (gloo through the host)"] = timed(exch)
    out["assembly + exchange"] = timed(lambda: (asm(), exch()))
    # one rank at a time
    for r in range(world):
        if r == rank:
            for _ in range(2): asm()
            t0 = time.perf_counter()
            for _ in range(5): asm()
            out["assembly, other rank idle"] = (time.perf_counter() - t0) / 5 * 1e3
        dist.barrier()
    q.put(out)
    dist.barrier(); D.close(); dist.destroy_process_group()


def main():
    from goldfish_amd import geometry as G, sharding
    spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
    gp = np.array([p.nel[0] * p.nel[1] * 16 for p in spec.patches])
    for world in (1, 2, 4, 8):
        part = sharding.partition_patches(spec, world)
        rows = []
        for r in range(world):
            mine = set(np.flatnonzero(part == r))
            mv = sum(i.npts for i in spec.interfaces if i.a in mine or i.b in mine)
            rows.append((int(gp[list(mine)].sum()), mv))
        tot_mv = sum(i.npts for i in spec.interfaces)
        print("%d ranks: Gauss points per rank max %d (%.3f of the model), mortar vertices per rank max %d (%.3f of the model's %d: cut interfaces are evaluated on both owners)"
              % (world, max(r[0] for r in rows), max(r[0] for r in rows) / gp.sum(), max(r[1] for r in rows), max(r[1] for r in rows) / tot_mv, tot_mv))
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda d: d["rank"])
    for p in procs: p.join(timeout=300)
    for d in res:
        print("rank %d (%d Gauss points, %d mortar vertices): " % (d["rank"], d["gauss_points"], d["mortar_points"]) +
              "; ".join("%s %.2f ms" % (k, v) for k, v in d.items() if k not in ("rank", "gauss_points", "mortar_points")))


if __name__ == "__main__":
    main()
