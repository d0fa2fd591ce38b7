"""profiles/{conv,act}_hbm_bytes_per_launch.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
of `bench.py --steps 1 --warmup 1`.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
counters are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled.
The file records a fingerprint of the kernel sources and the launch planner it was measured on: bench.py reports
`traffic: null` (with the reason) when the running sources differ.
usage: make_traffic_json.py <fetch_dir> <write_dir> <out.json> [conv|act]"""
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path


def source_fingerprint(root):
    """sha256 (12 hex digits) over everything that decides which launches a step makes and what they do: the HIP sources,
    the C ABI header and the host modules that pack weights and plan launches."""
    root = Path(root)
    files = sorted(f for f in (root / "flowhigh_amd" / "csrc").glob("*") if f.suffix in (".hip", ".h")) + [root / "include" / "flowhigh_hip.h"] + \
        [root / "flowhigh_amd" / f for f in ("packing.py", "planner.py", "runtime.py", "vocoder.py", "flow.py", "frontend.py")]
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    which = sys.argv[4] if len(sys.argv) > 4 else "conv"
    names = {"conv": ("conv_mfma_kernel", "conv_wino_kernel", "conv_wino54_kernel", "amp_actconv_kernel", "narrow_bf_kernel"),
             "act": ("act1d_strip_kernel",)}[which]

    def total(d, name):
        f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and any(k in r['Kernel_Name'] for k in names):
                tot += float(r['Counter_Value'])
                n += 1
        return tot, n
    fs, n1 = total(fetch_dir, 'FETCH_SIZE')
    ws, n2 = total(write_dir, 'WRITE_SIZE')
    assert n1 == n2 and n1 > 0
    per = (2.0 * fs + ws) * 1024.0 / n1
    json.dump({"bytes_per_launch": round(per), "launches": n1, "fetch_kb_raw": fs, "write_kb": ws,
               "read_bytes_per_launch": round(2.0 * fs * 1024.0 / n1), "write_bytes_per_launch": round(ws * 1024.0 / n1),
               "source_fingerprint": source_fingerprint(Path(__file__).resolve().parents[1]),
               "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches, " + " + ".join(names) + " dispatches of "
                          "`bench.py --steps 1 --warmup 1` (B = 1, 10 s clip)"},
              open(out, 'w'), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
