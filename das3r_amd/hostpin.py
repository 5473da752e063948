"""Host-side placement of a one-process-per-GPU worker (bench.py, das3r_amd.farm)."""
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _parse_cpulist(text):
    """'0-7,64-71' -> [0..7, 64..71]"""
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs_root="/"):
    """NUMA node of every GPU in HIP's enumeration order, from sysfs alone (no HIP call: the pin has to precede the runtime's
    first thread).  KFD topology nodes in numeric order that have SIMDs are the GPUs; their drm_render_minor names
    /sys/class/drm/renderD<minor>/device/numa_node.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (index lists) are applied.
    -> list of ints (-1 = unknown), [] when the topology is not there."""
    top = os.path.join(sysfs_root, "sys/class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(top) if d.isdigit())
    except OSError:
        return []
    nodes = []
    for i in ids:
        props = {}
        for line in (_read(os.path.join(top, str(i), "properties")) or "").splitlines():
            k, _, v = line.partition(" ")
            props[k] = v.strip()
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue   # a CPU node
        minor = props.get("drm_render_minor")
        numa = _read(os.path.join(sysfs_root, f"sys/class/drm/renderD{minor}/device/numa_node")) if minor else None
        try:
            nodes.append(int(numa))
        except (TypeError, ValueError):
            nodes.append(-1)
    # (the runtime applies ROCR's filter first, then HIP's; CUDA_VISIBLE_DEVICES is HIP's alias on ROCm — HIP and torch honour it
    #  when HIP_VISIBLE_DEVICES is not set)
    hip_var = "HIP_VISIBLE_DEVICES" if os.environ.get("HIP_VISIBLE_DEVICES") else "CUDA_VISIBLE_DEVICES"
    for var in ("ROCR_VISIBLE_DEVICES", hip_var):
        sel = os.environ.get(var)
        if sel:
            try:
                nodes = [nodes[int(t)] for t in sel.split(",") if t.strip() != ""]
            except (ValueError, IndexError):
                pass   # UUIDs or stale indices: keep the unfiltered order rather than guess
    return nodes


def choose_cpus(local_rank, allowed, numa_of_gpu=None, cpus_of_node=None):
    """The eight CPUs rank `local_rank` is pinned to.  allowed: sorted CPU ids this process may use.  With the topology known
    (numa_of_gpu[r] = NUMA node of GPU r, cpus_of_node[n] = CPU ids of node n) the complex is taken on the GPU's OWN node, the
    ranks whose GPUs share a node spreading over that node's complexes; without it rank r takes the 2r-th complex of the first
    SMT threads (eight ranks then spread over both sockets of a two-socket box)."""
    r = int(local_rank)

    def complexes(cpus):
        cpus = sorted(cpus)
        groups = [cpus[i:i + 8] for i in range(0, len(cpus) - 7, 8)]
        first = max(1, len(groups) // 2)     # Linux lists the first SMT thread of every core first
        return groups[:first] if groups else []

    if numa_of_gpu and cpus_of_node and r < len(numa_of_gpu) and numa_of_gpu[r] in cpus_of_node:
        node = numa_of_gpu[r]
        local = complexes([c for c in cpus_of_node[node] if c in set(allowed)])
        if local:
            peers = [i for i, n in enumerate(numa_of_gpu) if n == node]          # the GPUs on this node, in rank order
            k = peers.index(r)
            return local[(k * max(1, len(local) // max(len(peers), 1))) % len(local)]
    groups = complexes(allowed)
    if not groups:
        return None
    return groups[(r * max(1, len(groups) // 8)) % len(groups)]


def pin_to_ccx(local_rank, sysfs_root="/"):
    """Pin this process's threads-to-be (Python, the autograd engine's worker, the HIP runtime's helpers) to the eight CPUs of
    one core complex.  A forward + backward of a small scene is a ping-pong between those threads; when the scheduler spreads
    them over core complexes or sockets every hand-off pays a cross-fabric wake-up: measured 0.256 vs 0.197 ms per step on the
    100 k-splat 1080p workload (390 vs 508 Msplats/s on a 2 x 64-core EPYC 9575F), bimodal from run to run without the pin.
    The complex is taken on the NUMA node the rank's GPU hangs off (sysfs: gpu_numa_nodes) — on an 8-GPU node four GPUs sit on
    each socket, and a rank pinned to the far socket pays the fabric on every launch and on the pinned mailbox; when sysfs does
    not say, rank r takes the 2r-th complex.  Call it before the first HIP call (the runtime's threads inherit the mask) and
    after any host-side set-up that should keep every core (torch's CPU thread pool).  DAS3R_PIN=0 switches it off.
    Returns (previous mask, pinned CPUs) or None."""
    if os.environ.get("DAS3R_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    numa = gpu_numa_nodes(sysfs_root)
    cpus_of_node = {}
    for n in set(numa):
        if n >= 0:
            cpus = _parse_cpulist(_read(os.path.join(sysfs_root, f"sys/devices/system/node/node{n}/cpulist")))
            if cpus:
                cpus_of_node[n] = cpus
    mine = choose_cpus(local_rank, allowed, numa, cpus_of_node)
    if not mine:
        return None
    os.sched_setaffinity(0, mine)
    return allowed, mine


def worker_cpus(index, count, allowed):
    """The CPUs worker thread `index` of `count` (farm.run_jobs: K jobs in flight on one rank) runs on: ONE core of the rank's
    complex each, distinct — worker i takes allowed[i].  The rank's other threads (the HIP runtime's helpers, which inherit the whole
    complex) keep the remaining cores.  Two interpreter threads left to the scheduler inside one eight-core mask end up time-slicing
    one core whenever a runtime helper is busy on theirs; each on a core of its own, a job's launch loop is never descheduled for its
    neighbour's.  Needs at least two cores per worker (else -> None: no pin)."""
    allowed = sorted(allowed)
    if count < 1 or index < 0 or index >= count or len(allowed) < 2 * count:
        return None
    return [allowed[index]]


def pin_worker_thread(index, count):
    """Pin the CALLING thread (sched_setaffinity(0, ...) acts on the calling thread on Linux) to worker_cpus() of the process's mask
    at the time of the call — i.e. inside the complex pin_to_ccx chose for the rank.  DAS3R_PIN=0 switches it off.  -> the CPUs or None."""
    if os.environ.get("DAS3R_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        mine = worker_cpus(index, count, os.sched_getaffinity(0))
        if mine:
            os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


def unpin(pinned):
    """Give the process its previous CPU mask back (e.g. before a multi-threaded CPU baseline)."""
    if pinned:
        os.sched_setaffinity(0, pinned[0])
