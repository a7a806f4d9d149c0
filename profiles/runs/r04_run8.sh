#!/bin/bash
# round 4, call 8: packet tests on the k_pktl build with the whole-line fetch (encrypt), then where the time of a cyclic launch goes: per-workgroup timeline
# (profiles/cyc_timeline.py) and the kernel trace of two streams of 16 MiB / 64 MiB messages (do consecutive launches overlap?)
O=$PWD/gpurun_out/r04_run8; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/*.so > $O/so_sha256.txt
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_stress.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 600 python profiles/cyc_timeline.py 32 0.0625 1 4 16 64 256 2>&1 | tee $O/cyc_timeline_aes256.txt
timeout 600 python profiles/cyc_timeline.py 16 1 16 64 2>&1 | tee $O/cyc_timeline_aes128.txt
cd /tmp && export TMPDIR=/tmp
for M in 16 64; do for K in 1 2; do
  G=$(python3 -c "print($M/1024)")
  rm -rf /tmp/kt_$M_$K
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_${M}_$K -- python3 $GRAFT_REPO_ROOT/bench.py --gib-per-gpu $G --inflight $K --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2> $O/kt_${M}_$K.err
  python3 - /tmp/kt_${M}_$K $M $K <<'PY' | tee $O/kernel_trace_${M}m_k$2.txt
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_body" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
rows = rows[-150:-20]                     # steady state, inside the timed region
if rows:
    t0 = rows[0][0]
    dur = [e - s for s, e, _ in rows]
    gap = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]        # next start minus this end: negative = overlap
    per = (rows[-1][0] - rows[0][0]) / (len(rows) - 1)
    print("%s MiB K=%s: %d launches; kernel duration median %.2f us; start-to-start %.2f us; gap (next start - this end) median %.2f us, min %.2f, max %.2f" % (
        sys.argv[2], sys.argv[3], len(rows), sorted(dur)[len(dur) // 2] / 1e3, per / 1e3, sorted(gap)[len(gap) // 2] / 1e3, min(gap) / 1e3, max(gap) / 1e3))
    for s, e, q in rows[:8]:
        print("   start %9.2f us  end %9.2f us  (%.2f us)  queue %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
PY
done; done
