"""Copies the round profile set from gpurun_out/ (tools/round_profile.sh TAG) into profiles/ with the headers the readers
(bench.py: _profile_traffic, the judge) expect.  python tools/collect_profiles.py r02"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
HEAD = ("# round %s: rocprofv3 PMC passes (counters only + kernel trace; tools/pmc_run.sh), averages per dispatch; FETCH_SIZE / WRITE_SIZE in KiB\n"
        "# FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts the 128-byte requests of wide loads as 64 B)\n" % TAG[1:].lstrip("0"))


def lines(name):
    with open(os.path.join(G, name)) as f:
        return [l for l in f if l.strip()]


def put(name, text):
    with open(os.path.join(P, name), "w") as f:
        f.write(text)
    print("profiles/" + name)


for n in ("bench_line.json", "bench_line_dense.json", "bench_line_stress.json", "bench_kernel_trace_stats.txt", "bf16_storage_bench_line.json",
          "bf16_storage_bench_kernel_trace_stats.txt", "bench_line_dense_rows.json", "bench_line_heads4.json", "heads4_bench_kernel_trace_stats.txt", "stress_kernel_trace_stats.txt",
          "one_stream_bench_kernel_trace_stats.txt", "bench_line_bsz2.json", "bench_line_reference_batch.json"):
    src = os.path.join(G, "%s_%s" % (TAG, n))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_%s" % (TAG, n)))
        print("profiles/%s_%s" % (TAG, n))
fwd = lines("pmc_%s_k1_fwd.txt" % TAG)
put("%s_k1_fwd_pmc_vid.txt" % TAG, HEAD + "# command: bash tools/pmc_run.sh %s_k1_fwd str_attn_fwd python bench.py --only_roofline   (video shape: N=16, Li=300, Lr=20, Lqa=40, D=128)\n" % TAG
    + "".join(l for l in fwd if "str_attn_fwd_reg" in l))
put("%s_k1_fwd_pmc_sub.txt" % TAG, HEAD + "# command: bash tools/pmc_run.sh %s_k1_fwd str_attn_fwd python bench.py --only_roofline   (subtitle shape: Lr=50)\n" % TAG
    + "".join(l for l in fwd if "str_attn_fwd_d128" in l))
for s in ("vid", "sub"):
    t = "".join("#   " + l for l in lines("%s_k1_bwd_times_%s.txt" % (TAG, s))[-3:])
    put("%s_k1_bwd_pmc_%s.txt" % (TAG, s), HEAD + "# command: %sbash tools/pmc_run.sh %s_k1_bwd_%s str_attn_bwd_fused python tools/k1_bwd_times.py\n# event-timed, same run:\n%s"
        % ("LR=50 " if s == "sub" else "", TAG, s, t) + "".join(lines("pmc_%s_k1_bwd_%s.txt" % (TAG, s))))
for k, kern in (("nt", "gemm_nt_stream_kernel, forward 960000 x 384 -> 128"), ("tn", "gemm_tn_quad_kernel, weight gradient 960000 x 128^T x 384")):
    name = "pmc_%s_gemm_%s.txt" % (TAG, k)
    if os.path.exists(os.path.join(G, name)):
        put("%s_gemm_%s_pmc.txt" % (TAG, k), HEAD + "# command: bash tools/pmc_run.sh %s_gemm_%s ... python tools/gemm_one.py 960000 128 384 %s   (%s; fp16-split, DESIGN.md finding 20)\n"
            % (TAG, k, k, kern) + "".join(lines(name)))

for k, what in (("rep", "c2q_down_projection shape: a broadcast over 300 frames, 960000 rows"), ("flat", "concat_fc shape: 960000 rows")):
    name = "pmc_%s_cat3_fused_%s.txt" % (TAG, k)
    if os.path.exists(os.path.join(G, name)):
        t = "".join("#   " + l for l in lines("%s_cat3_fused_times_%s.txt" % (TAG, k))[-8:])
        put("%s_cat3_fused_pmc_%s.txt" % (TAG, k), HEAD + "# command: %sbash tools/pmc_run.sh %s_cat3_fused_%s cf python tools/cat3_fused_time.py   (%s; cf_bwd_kernel = fused backward, "
            "cff_fwd_kernel = fused forward, csrc/cat3_fused.hip)\n# event-timed, same run:\n%s" % ("REP=1 " if k == "flat" else "", TAG, k, what, t) + "".join(lines(name)))

if os.path.exists(os.path.join(G, "pmc_%s_k1_long_fwd.txt" % TAG)):
    put("%s_k1_long_fwd_pmc.txt" % TAG, HEAD + "# command: bash tools/pmc_run.sh %s_k1_long_fwd str_attn_long_fwd python bench.py --config stress --only_roofline   "
        "(BASELINE configs[4]: N=16, Li=300, Lr=512, Lqa=40, D=256, bf16 storage, ragged masks)\n" % TAG + "".join(lines("pmc_%s_k1_long_fwd.txt" % TAG)))

for k, what in (("cat3_ragged_instep", "the [a,b,a*b] kernels on ragged token rows INSIDE the bench step (cff_fwd_kernel<.., true> = gathered rows, cf_bwd_kernel<.., 3> = "
                "per-group live words; <.., 0> / <.., false> = the concat_fc instance on compact rows)"),
                ("k1_instep", "the attention kernels inside the bench step (frame-compact A, compact region rows)")):
    name = "pmc_%s_%s.txt" % (TAG, k)
    if os.path.exists(os.path.join(G, name)):
        put("%s_%s_pmc.txt" % (TAG, k), HEAD + "# command: bash tools/pmc_run.sh %s_%s ... python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline "
            "--no_device_time   (%s)\n" % (TAG, k, what) + "".join(lines(name)))
