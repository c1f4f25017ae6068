"""Fold the rocprofv3 --pmc CSVs written by tools/collect_profiles.sh into one JSON (per-kernel,
per-launch counter sums plus the derived figures DESIGN.md quotes).
Usage: make_pmc_json.py <dir> <tag> [mode genomes length [staging]]   (defaults: the bench.py N=1 workload)"""
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
MODE = sys.argv[3] if len(sys.argv) > 3 else "minhash"
G = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
L = int(sys.argv[5]) if len(sys.argv) > 5 else 5_000_000
STAGING = sys.argv[6] if len(sys.argv) > 6 else None  # "packed": bench.py --staging packed
K, S = 21, 1000
want = ("synth_kernel", "sketch_minhash_kernel", "sketch_minhash_packed_kernel", "sketch_kssd", "transpose_slices_kernel", "pair_tiled_kernel", "pair_join_phase")
# pair_join_phase = every kernel of the inverted join: rocPRIM sort / scan / reduce + join_* (and the forest's two rocPRIM sorts).
# The library's rocPRIM is /opt/rocm's (inline namespace ROCPRIM_400200_NS); torch carries its own (ROCPRIM_400001_NS: the
# partition / scan kernels of bench.py's packing of a batch with torch ops are NOT part of the phase).
JOIN_PARTS = ("ROCPRIM_400200_NS", "join_")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = next((w for w in want if w in r["Kernel_Name"]), None)
        if name is None and any(p in r["Kernel_Name"] for p in JOIN_PARTS):
            name = "pair_join_phase"
        if name is None:
            continue
        if MODE == "dense_pairs" and name == "pair_tiled_kernel" and "pair_tiled_kernel<unsigned int" in r["Kernel_Name"]:
            continue  # the u32 toy set that warms the code path in front of the first call on the real (u64) set
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        # the gated second launch over the partial segments (runtime-k instantiation, every workgroup leaves at once unless the
        # merge flagged its genome) belongs to the sketch call of the compile-time-k launch in front of it: no launch of its own
        if name.startswith("sketch_minhash") and "_kernel<0," in r["Kernel_Name"]:
            continue
        # the join is many kernels per pair phase: its "launch" is the phase (one per bench step, --steps 1 in the PMC runs)
        disp[name][r["Counter_Name"]].add("phase" if name == "pair_join_phase" else r["Dispatch_Id"])
out = {
    "command": "rocprofv3 --pmc <group> --output-format csv -- python bench.py [--mode kssd] --steps 1 --warmup 0 "
               "--no-cpu-baseline (one run per counter group, tools/collect_profiles.sh)",
    "workload": {"genomes": G, "length": L, "k": K, "s": S, "mode": MODE, **({"staging": STAGING} if STAGING else {})},
    "units": "hbm_bytes_per_launch = fabric-side bytes = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*WRREQ_64B + "
             "32*(WRREQ - WRREQ_64B), Infinity-Cache hits included (MI355X_MICROARCH.md, HBM section).  FETCH_SIZE / "
             "WRITE_SIZE (1024-byte units) are kept beside them: FETCH_SIZE = RDREQ x 64 B, i.e. half the bytes of "
             "128-byte requests -- the guide's gfx950 'x2' correction; WRITE_SIZE is checked on synth_kernel, which "
             "writes exactly genomes*length bytes.  SQ_* are sums over the chip.",
    "kernels": {},
}
for name in want:
    if name not in tot:
        continue
    if STAGING and name == "pair_join_phase":
        continue  # the packing of the batch (torch plumbing outside the timed region) runs rocPRIM kernels of its own: no clean figure here
    k = {}
    for c, v in tot[name].items():
        k[c + "_per_launch"] = v / max(len(disp[name][c]), 1)
    # Exact fabric-side bytes from the request-size counters (every read request of these kernels is
    # a 128-byte one; FETCH_SIZE tallies each request at 64 B, hence the guide's "x2" for gfx950).
    rd = (128.0 * k.get("TCC_EA0_RDREQ_128B_sum_per_launch", 0) + 64.0 * k.get("TCC_EA0_RDREQ_64B_sum_per_launch", 0)
          + 32.0 * k.get("TCC_EA0_RDREQ_32B_sum_per_launch", 0))
    w64 = k.get("TCC_EA0_WRREQ_64B_sum_per_launch", 0)
    wr = 64.0 * w64 + 32.0 * max(k.get("TCC_EA0_WRREQ_sum_per_launch", 0) - w64, 0)
    if rd > 0:
        k["fabric_read_bytes_per_launch"] = rd
        k["fabric_write_bytes_per_launch"] = wr
        k["hbm_bytes_per_launch"] = rd + wr
    elif "FETCH_SIZE_per_launch" in k and "WRITE_SIZE_per_launch" in k:
        k["hbm_bytes_per_launch"] = (2.0 * k["FETCH_SIZE_per_launch"] + k["WRITE_SIZE_per_launch"]) * 1024.0
    if name in ("sketch_minhash_kernel", "sketch_minhash_packed_kernel", "sketch_kssd") and "SQ_INSTS_VALU_per_launch" in k:
        steps = G * L / 64.0
        k["derived"] = {
            "kmer_wave_steps": steps,
            "valu_insts_per_step": k["SQ_INSTS_VALU_per_launch"] / steps,
            "salu_insts_per_step": k.get("SQ_INSTS_SALU_per_launch", 0) / steps,
            "lds_insts_per_step": k.get("SQ_INSTS_LDS_per_launch", 0) / steps,
            "wave_cycles_per_step": k.get("SQ_WAVE_CYCLES_per_launch", 0) / steps,
        }
    out["kernels"][name] = k
if "synth_kernel" in out["kernels"] and "WRITE_SIZE_per_launch" in out["kernels"]["synth_kernel"]:
    out["write_size_calibration"] = {"synth_bytes_written": float(G) * L,
                                     "WRITE_SIZE": out["kernels"]["synth_kernel"]["WRITE_SIZE_per_launch"],
                                     "bytes_per_unit": float(G) * L / out["kernels"]["synth_kernel"]["WRITE_SIZE_per_launch"]}
print(json.dumps(out, indent=1))
