"""Host placement of the one-process-per-GPU workers (das3r_amd/hostpin.py): the core complex is taken on the NUMA node the
rank's GPU hangs off, read from sysfs without a HIP call.  A fake sysfs tree stands in for an 8-GPU node."""
import os

from das3r_amd import hostpin


def _fake_node(root, gpus_numa, cpus_per_node=64, sockets=2):
    """KFD topology: `sockets` CPU nodes first, then one node per GPU; render minors 128 + i."""
    top = root / "sys/class/kfd/kfd/topology/nodes"
    for s in range(sockets):
        (top / str(s)).mkdir(parents=True)
        (top / str(s) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        nd = root / f"sys/devices/system/node/node{s}"
        nd.mkdir(parents=True)
        first = f"{s * cpus_per_node}-{(s + 1) * cpus_per_node - 1}"
        second = f"{sockets * cpus_per_node + s * cpus_per_node}-{sockets * cpus_per_node + (s + 1) * cpus_per_node - 1}"   # SMT siblings
        (nd / "cpulist").write_text(first + "," + second + "\n")
    for i, numa in enumerate(gpus_numa):
        n = top / str(sockets + i)
        n.mkdir(parents=True)
        (n / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + i}\n")
        dev = root / f"sys/class/drm/renderD{128 + i}/device"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{numa}\n")


def test_gpu_numa_nodes_from_sysfs(tmp_path, monkeypatch):
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    _fake_node(tmp_path, [0, 0, 0, 0, 1, 1, 1, 1])
    assert hostpin.gpu_numa_nodes(str(tmp_path)) == [0, 0, 0, 0, 1, 1, 1, 1]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,2")
    assert hostpin.gpu_numa_nodes(str(tmp_path)) == [1, 0]
    assert hostpin.gpu_numa_nodes(str(tmp_path / "nothing_here")) == []


def test_every_rank_lands_on_its_gpus_node_and_on_its_own_complex():
    allowed = list(range(256))                                      # 2 sockets x 64 cores x 2 threads
    cpus_of_node = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    # GPUs interleaved over the sockets on purpose: the old rule (rank r -> complex 2r) would put ranks 1, 3 on the wrong socket
    numa = [0, 1, 0, 1, 0, 1, 0, 1]
    picks = [hostpin.choose_cpus(r, allowed, numa, cpus_of_node) for r in range(8)]
    for r, cpus in enumerate(picks):
        assert len(cpus) == 8 and set(cpus) <= set(cpus_of_node[numa[r]]), (r, cpus)
        assert max(cpus) < 128, "first SMT threads only"
    assert len({tuple(c) for c in picks}) == 8, "no two ranks share a core complex"
    # a restricted affinity mask (container with 32 CPUs of node 1) is respected
    few = list(range(64, 96))
    c = hostpin.choose_cpus(1, few, numa, cpus_of_node)
    assert set(c) <= set(few)


def test_fallback_without_topology_is_the_old_rule():
    allowed = list(range(256))
    assert hostpin.choose_cpus(0, allowed) == list(range(0, 8))
    assert hostpin.choose_cpus(3, allowed) == list(range(48, 56))    # rank r -> the 2r-th complex of the first SMT threads
    assert hostpin.choose_cpus(0, list(range(4))) is None            # fewer than eight CPUs: no pin


def test_pin_to_ccx_applies_a_mask(tmp_path, monkeypatch):
    monkeypatch.delenv("DAS3R_PIN", raising=False)
    before = os.sched_getaffinity(0)
    try:
        got = hostpin.pin_to_ccx(0, sysfs_root=str(tmp_path))       # no topology there: fallback rule on the CPUs this process has
        if len(before) >= 8:
            assert got is not None and set(got[1]) <= before and os.sched_getaffinity(0) == set(got[1])
        else:
            assert got is None
    finally:
        os.sched_setaffinity(0, before)


def test_worker_threads_get_distinct_cores_of_the_ranks_mask():
    """K jobs in flight on one rank — every worker thread on a core of its own inside the rank's complex."""
    import os
    import threading
    from das3r_amd import farm, hostpin
    assert hostpin.worker_cpus(0, 2, range(8, 16)) == [8] and hostpin.worker_cpus(1, 2, range(8, 16)) == [9]
    assert hostpin.worker_cpus(2, 3, [3, 1, 2, 0, 4, 5]) == [2]
    assert hostpin.worker_cpus(0, 2, [0, 1, 2]) is None and hostpin.worker_cpus(2, 2, range(8)) is None   # too few cores / no such worker
    mask = sorted(os.sched_getaffinity(0))
    if len(mask) < 4:
        return
    seen, gate = {}, threading.Barrier(2)

    def job(i):
        gate.wait(10)                       # both workers are alive at once: two distinct threads
        seen[i] = (threading.get_ident(), sorted(os.sched_getaffinity(0)))
        return i

    assert farm.run_jobs(range(2), job, 2, None, pin=True) == [0, 1]
    (t0, c0), (t1, c1) = seen[0], seen[1]
    assert t0 != t1 and len(c0) == len(c1) == 1 and c0 != c1 and set(c0 + c1) <= set(mask), seen
    assert sorted(os.sched_getaffinity(0)) == mask, "the caller's own mask is untouched"
    seen.clear()
    gate = threading.Barrier(2)
    farm.run_jobs(range(2), job, 2, None)   # default on the host: no pin
    assert seen[0][1] == mask and seen[1][1] == mask
