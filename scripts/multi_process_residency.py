#!/usr/bin/env python3
"""Several processes evaluating on ONE GPU at the same time: does the single-launch kernel's residency requirement hold up?

    python scripts/multi_process_residency.py <processes> <n_molecular> <evaluations> [persistent] [verify]

"verify": every evaluation writes into a force array poisoned with NaN just before and is compared bit for bit with the
first one on the device (two extra small kernels per step, no host synchronisation) -- that is how the one starved
evaluation, whichever it is, gets checked.

The parent never touches the GPU; it starts <processes> children (fresh interpreters), which build their own system,
wait for a common start time and then evaluate back to back, reading the result every 25 evaluations (that is where a
failed evaluation would surface as CAVMD_ERR_SYNC_TIMEOUT).  Each child reports its rate, the errors it saw, whether it met
a starved evaluation ("sync_timeout_seen") and whether its single launch is suspended at the end.  Keep <processes> <= 6 (gpurun's process guard).
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def child(n_mol, evals, persistent, start_at, verify):
    sys.path.insert(0, os.path.join(ROOT, "cav-hoomd_amd"))
    import numpy as np
    import torch
    import cavitymd
    from cavitymd import _capi, synthetic

    cfg = synthetic.config3(seed=1 + os.getpid() % 7, n_molecular=n_mol)
    n = len(cfg["charge"])
    dev = "cuda"
    pos = torch.from_numpy(np.concatenate([cfg["position"], cavitymd.state.type_tag_as_double(cfg["typeid"])[:, None]],
                                          axis=1)).to(dev)
    chg = torch.from_numpy(cfg["charge"]).to(dev)
    img = torch.from_numpy(cfg["image"]).to(dev)
    frc = torch.empty((n, 4), dtype=torch.float64, device=dev)
    p = cfg["params"]
    prm = _capi.make_params(p["omegac"], p["couplstr"], p["phmass"])
    ws = _capi.Workspace(n)
    if persistent is not None:
        ws.set_tunable("persistent", persistent)
    L_typeid = cfg["types"].index("L")
    args = (0, n, pos.data_ptr(), chg.data_ptr(), img.data_ptr(), cfg["box"], L_typeid, prm, frc.data_ptr())
    ws.compute_hoomd(*args)
    want = np.array(ws.result().dipole[:])
    torch.cuda.synchronize()
    want_f = frc.clone()
    force_mismatches = torch.zeros((), dtype=torch.int64, device=dev)
    while time.time() < start_at:
        time.sleep(0.001)
    timeouts = wrong = done = 0
    t0 = time.perf_counter()
    for it in range(evals):
        try:
            if verify:
                frc.fill_(float("nan"))
            ws.compute_hoomd(*args)
            if verify:
                force_mismatches += (frc.view(torch.int64) != want_f.view(torch.int64)).any()
            done += 1
            if it % 25 == 24:
                wrong += int(not np.array_equal(np.array(ws.result().dipole[:]), want))
        except _capi.CavmdError as e:
            if e.status != _capi.CAVMD_ERR_SYNC_TIMEOUT:
                raise
            timeouts += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"pid": os.getpid(), "N": n, "evals_per_s": done / dt, "timeouts": timeouts, "wrong_results": wrong,
                      "evaluations_with_wrong_forces": int(force_mismatches.item()) if verify else None,
                      "single_launch_suspended": ws.get_tunable("persistent_suspended"),
                      "sync_timeout_seen": ws.get_tunable("sync_timeout_seen")}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), None if sys.argv[4] == "auto" else int(sys.argv[4]), float(sys.argv[5]),
              sys.argv[6] == "verify")
        return
    procs, n_mol, evals = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    persistent = sys.argv[4] if len(sys.argv) > 4 else "auto"
    verify = "verify" if len(sys.argv) > 5 and sys.argv[5] == "verify" else "plain"
    if procs > 6:
        raise SystemExit("at most 6 processes on the card")
    start_at = time.time() + 60.0 + 20.0 * procs  # children need to import torch and build their systems first
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(n_mol), str(evals), persistent,
                              repr(start_at), verify]) for _ in range(procs)]
    rc = 0
    deadline = time.time() + 600
    for k in kids:
        try:
            rc |= k.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            k.kill()
            rc |= 1
    print(f"# {procs} processes, n_molecular={n_mol}, {evals} evaluations each, persistent={persistent}, {verify}: rc={rc}")
    sys.exit(rc)


if __name__ == "__main__":
    main()
