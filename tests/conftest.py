import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _cap_thread_pools():
    """Before numpy / torch / the OpenMP oracle are imported: size their thread pools to the CPUs this process can keep busy
    (affinity mask capped by the cgroup's CPU quota).  On the GPU boxes 256 logical CPUs are visible under a 16-CPU quota; pools
    sized from the visible CPUs spin-wait after each parallel region, exhaust the quota and get EVERY thread of the container
    frozen for up to 100 ms (ei-nexus_official_amd/placement.py::cgroup_cpu_quota, profiles/r05_notes.md)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("einx_placement", os.path.join(ROOT, "ei-nexus_official_amd", "placement.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.cap_thread_pools()


HOST_THREADS = _cap_thread_pools()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _gpu_tests_need_a_device(request):
    """every -m gpu test: a HIP device must be there (no CPU fallback to pass on), and the device is idle again afterwards"""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    assert torch.cuda.is_available(), "these tests need a HIP device"
    yield
    torch.cuda.synchronize()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


def pytest_sessionfinish(session, exitstatus):
    """Measured float errors of the tolerance-based comparisons (helpers.close_and_record): printed and kept as
    gpurun_out/parity_errors.json (the GPU box merges gpurun_out/ back)."""
    import json
    from helpers import recorded_errors, recorded_flips
    errs = recorded_errors()
    flips = recorded_flips()
    if not errs and not flips:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_errors.json"), "w") as f:
            json.dump(dict(errs, **({"assignment_flips": flips} if flips else {})), f, indent=1, sort_keys=True)
    except OSError:
        pass
    tr = session.config.pluginmanager.get_plugin("terminalreporter")
    if tr is not None:
        tr.write_line("")
        tr.write_line("measured max |error| of the tolerance-based comparisons (tag: measured / atol, largest reference magnitude):")
        for k in sorted(errs):
            e = errs[k]
            tr.write_line(f"  {k}: {e['max_abs_err']:.3e} / {e['atol']:.1e}   (|ref| <= {e['max_abs_ref']:.3g}, {e['n']} values)")
        if flips:
            tr.write_line("match-assignment flips (tag: flips / assignments compared, matched in the checker; margins of flipped rows):")
            for k in sorted(flips):
                e = flips[k]
                tr.write_line(f"  {k}: {e['flips']} / {e['compared']} ({e['matched']} matched)" + (f"  margins {e['margins']}" if e["margins"] else ""))
