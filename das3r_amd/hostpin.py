"""Host-side placement of a one-process-per-GPU worker (bench.py, das3r_amd.farm)."""
import os


def pin_to_ccx(local_rank):
    """Pin this process's threads-to-be (Python, the autograd engine's worker, the HIP runtime's helpers) to the eight CPUs of
    one core complex.  A forward + backward of a small scene is a ping-pong between those threads; when the scheduler spreads
    them over core complexes or sockets every hand-off pays a cross-fabric wake-up: measured 0.256 vs 0.197 ms per step on the
    100 k-splat 1080p workload (390 vs 508 Msplats/s on a 2 x 64-core EPYC 9575F), bimodal from run to run without the pin.
    Rank r takes the 2r-th complex, so eight ranks spread over both sockets.  Call it before the first HIP call (the runtime's
    threads inherit the mask) and after any host-side set-up that should keep every core (torch's CPU thread pool).
    DAS3R_PIN=0 switches it off.  Returns (previous mask, pinned CPUs) or None."""
    if os.environ.get("DAS3R_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    groups = [allowed[i:i + 8] for i in range(0, len(allowed) - 7, 8)]
    if not groups:
        return None
    first_threads = max(1, len(groups) // 2)     # Linux lists the first SMT thread of every core first
    mine = groups[(int(local_rank) * max(1, first_threads // 8)) % first_threads]
    os.sched_setaffinity(0, mine)
    return allowed, mine


def unpin(pinned):
    """Give the process its previous CPU mask back (e.g. before a multi-threaded CPU baseline)."""
    if pinned:
        os.sched_setaffinity(0, pinned[0])
