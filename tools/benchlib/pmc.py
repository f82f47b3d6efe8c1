"""rocprofv3 PMC child passes of a bench.py run: FETCH_SIZE / WRITE_SIZE per launch of named kernels (separate passes, as
MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950)."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile


def pmc_traffic(args, kernel_substrings, bench_path):
    """
    Mean FETCH_SIZE / WRITE_SIZE per launch of the kernels whose name contains one of ``kernel_substrings``
    (dict label -> substring).  Two child runs (the two counters do not fit one pass).  Returns
    ({label: {"fetch": bytes, "write": bytes, "launches": n}}, note).
    """
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(bench_path), "--pmc-child", "--workload", args.workload, "--batch", str(args.batch),
             "--method", args.method, "--dtype", args.dtype, "--steps", "12", "--warmup", "3", "--spots", str(args.spots),
             "--sparse-columns", str(args.sparse_columns)]
    for o in args.opt:
        child += ["--opt", o]
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    out = {k: {"fetch": None, "write": None, "launches": 0} for k in kernel_substrings}
    names = {}
    for counter, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        d = tempfile.mkdtemp(prefix="hgs_pmc_", dir="/tmp")
        try:
            p = subprocess.run([exe, "--output-format", "csv", "--pmc", counter, "-d", d, "-o", "pmc", "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {p.returncode}): " + p.stdout.decode(errors="replace")[-300:]
            acc = {k: [0.0, 0] for k in kernel_substrings}
            for path in files:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") != counter:
                            continue
                        kn = row.get("Kernel_Name", "")
                        for label, sub in kernel_substrings.items():
                            if sub in kn:
                                acc[label][0] += float(row.get("Counter_Value", 0) or 0)
                                acc[label][1] += 1
                                names[label] = kn
            for label, (tot, n) in acc.items():
                if n:
                    # rocprofv3 reports KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950 (x 2, guide)
                    out[label][key] = tot / n * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
                    out[label]["launches"] = n
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for label in out:
        out[label]["kernel_name"] = names.get(label)
    return out, ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this run (12 steps each); FETCH_SIZE x 2 "
                 "(gfx950 half-count), WRITE_SIZE as reported; the fabric counters include Infinity-Cache hits")
