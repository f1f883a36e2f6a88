"""Worker for the multi-process driver tests: one rank of a torch.distributed job running
the host driver with the CPU oracle injected as backend (tests only) or the HIP backend."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, backend_kind, N, npc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist
    import _mcs_loader
    mcs = _mcs_loader.load()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # "oracle": CPU oracle over gloo; "hip-gloo": HIP backend, all ranks on GPU 0, CUDA tensors over
    # gloo (rehearses the device-side merge on a one-GPU box); "hip": one GPU per rank over RCCL
    dist.init_process_group("nccl" if backend_kind == "hip" else "gloo", rank=rank, world_size=world)
    # further arguments: "2" (two species) and key=value options: itrs=, gather_max=, skew_max=, species_tallies=, finalize=
    two_species = "2" in sys.argv[5:]
    opt = dict(a.split("=", 1) for a in sys.argv[5:] if "=" in a)
    n_itrs = int(opt.get("itrs", 2))
    kw = {}
    if two_species:
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)], energy_transfer_frac=0.1)
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=n_itrs, **kw)
    prob = mcs.inputs.build_problem(cfg)
    if backend_kind == "oracle":
        import orc
        be = orc.OracleBackend(mcs.capi, "det", 1)
        dev = None
    else:
        from mcs_amd import hip_backend
        local = 0 if backend_kind == "hip-gloo" else int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        be = hip_backend.HipBackend(local, torch_tallies=True)
        dev = torch.device("cuda", local)
    be.create(prob)
    comm = mcs.driver.Comm(True, dev)
    res = mcs.driver.run(prob, be, comm, n_itrs=n_itrs, max_pcuts=npc, gather_max=int(opt.get("gather_max", 1 << 17)),
                         skew_max=float(opt.get("skew_max", 1.1)), species_tallies=opt.get("species_tallies", "full"),
                         finalize=opt.get("finalize", "0") == "1")
    if rank == 0:
        np.savez(out, f=res.tallies_f64, i=res.tallies_i64,
                 stats=np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in res.stats]),
                 n_use_max=np.array([s.n_use_max for s in res.stats]), split=np.array([s.split for s in res.stats]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
