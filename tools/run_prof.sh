cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a /tmp/prof_b
rocprofv3 --kernel-trace -d /tmp/prof_a -o tr -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --opt overlap_cameras=false > /root/repo/gpurun_out/prof_a.log 2>&1
DB=$(find /tmp/prof_a -name "*.db" | head -1)
python /root/repo/profiles/timeline.py $DB 46 > /root/repo/gpurun_out/timeline_noov.txt 2>&1
rocprofv3 --kernel-trace -d /tmp/prof_b -o tr -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /root/repo/gpurun_out/prof_b.log 2>&1
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python /root/repo/profiles/timeline.py $DB 44 > /root/repo/gpurun_out/timeline_ov.txt 2>&1
tail -2 /root/repo/gpurun_out/prof_b.log | cut -c1-300
