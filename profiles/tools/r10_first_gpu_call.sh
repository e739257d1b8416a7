#!/bin/bash
# The first GPU call of a round that had none: is HEAD green on an MI355X, and what does the driver's
# command print?   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/tools/r10_first_gpu_call.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r10_first; mkdir -p $OUT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.log 2>&1; grep '^{' $OUT/bench_full.log > $OUT/driver_full_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o r10 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-host-legs > $OUT/trace_driver.log 2>&1
grep '^{' $OUT/trace_driver.log > $OUT/driver_bench.json
find $OUT/trace_driver -name "*kernel_stats.csv" -exec cp {} $OUT/driver_kernel_stats.csv \;
# the link format's A/B: the same host legs in the round-5 format and with each round-6 part left out
for off in 31 24 8 16; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --link-flags-off $off 2>/dev/null | grep '^{' > $OUT/link_off_$off.json; done
tail -3 $OUT/pytest.log; tail -1 $OUT/smoke.log
python - <<'PY'
import json
for f in ("driver_full_line.json", "driver_bench.json", "link_off_31.json", "link_off_24.json", "link_off_8.json", "link_off_16.json"):
    try:
        j = json.loads(open("gpurun_out/r10_first/" + f).read().splitlines()[-1])
        d = j.get("delivered") or {}
        alt = d.get("with_read_values_on_the_link") or {}
        if alt: print(f, "delivered with read values on the link: %s cycles/s at %s B/cycle, bound by %s" % (alt.get("cycles_per_s"), alt.get("bytes_per_cycle"), alt.get("bound_by")))
        print(f, "value %.3g" % j["value"], "frac %.3f" % j["roofline"]["frac"], "kernel_ms %.4f" % j["kernel_ms"], "delivered %s B/cycle %s flags %s bound by %s, host replay %s" % (d.get("cycles_per_s"), d.get("bytes_per_cycle"), d.get("link_flags"), d.get("bound_by"), d.get("host_replay_cycles_per_s")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
