"""Rehearsal (not a measurement of scaling): the distributed factorisation with WORLD ranks sharing GPU 0 over gloo on a model of GF_REH_PATCHES^2 patches --
that it runs at size, what each phase of a rank takes while the other ranks use the same GPU, the memory per rank, the solution against the right-hand side."""
import os, sys, socket, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G


def worker(rank, world, port, n):
    import torch, torch.distributed as dist
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = G.synthetic_shell(n, n, nel=48, p=3, jitter=2)
    t = time.perf_counter(); nm = NonMatchingOpt.from_spec(spec, comm=dist, device=0); nm.sharded_solver = "distributed"
    nm.update_uIGA(G.smooth_displacement(spec, 0.5 * spec.h_th)); nm._assemble(3); t_setup = time.perf_counter() - t
    b = np.random.default_rng(1).standard_normal(nm.vec_iga_dof)
    t = time.perf_counter(); x = nm.solve_K(b); t_first = time.perf_counter() - t
    ds = nm._dsolver
    nm._k_version += 1                                  # as after a new assembly: numeric phase only
    t = time.perf_counter(); x = nm.solve_K(b); t_again = time.perf_counter() - t
    tm = dict(ds.timings)
    t = time.perf_counter(); x0 = ds.solve(b, max_refine=0); t_sub = time.perf_counter() - t
    r = b - nm.dev.apply(0, x)
    info = ds.info()
    free, tot = torch.cuda.mem_get_info()
    line = ("rank %d of %d: %d dofs, %d subtrees of %d (top: %d fronts); set-up + assembly %.1f s, first solve (symbolic phase, handles, factorisation) %.1f s; refactor + solve %.3f s "
            "= K values %.3f + own subtrees %.3f + Schur all-gather %.3f + top %.3f + solve with refinement; substitutions only %.3f s; |b - K x| / |b| %.1e; factor memory of this rank %.1f GB"
            % (rank, world, nm.vec_iga_dof, len(ds.my_roots), len(ds.roots), int((ds.owner == -1).sum()), t_setup, t_first, t_again, tm["k_values"], tm["own_subtrees"], tm["schur_allgather"],
               tm["top"], t_sub, np.linalg.norm(r) / np.linalg.norm(b), info["device_bytes"] / 1e9))
    for q in range(world):
        if q == rank: print(line, flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world, n = int(os.environ.get("GF_REH_WORLD", "2")), int(os.environ.get("GF_REH_PATCHES", "8"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, world, port, n)) for r in range(world)]
    [p.start() for p in ps]; [p.join() for p in ps]
    sys.exit(max(p.exitcode or 0 for p in ps))
