#!/bin/bash
# (round 6) The round's closing measurements on ONE box, from the repo root on the GPU box:  tools/profile_close.sh <tag f32> <tag bf16>
# GPU tests + smoke, rocprofv3 passes of the 720p fp32 forward and the 1080p B = 4 bf16 forward (tools/profile_passes.sh + profile_merge.py), the default
# bench line (parity of the timed graph, sensors, the real 720p CPU run, configs[2] / [4]), the bench command under rocprofv3 --kernel-trace --stats
# (the dominant kernel's average duration must agree with the line's live figure), and the side configurations.  Everything lands under gpurun_out/.
A=${1:-r06_x}; B=${2:-r06_y}
git rev-parse HEAD > gpurun_out/${A}_HEAD 2>/dev/null
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${A}_gputests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 >> gpurun_out/${A}_gputests.txt
tools/profile_passes.sh gpurun_out/$A > gpurun_out/${A}_passes.log 2>&1
python tools/profile_merge.py gpurun_out/$A gpurun_out/$A > gpurun_out/${A}_merge.log 2>&1
tools/profile_passes.sh gpurun_out/$B --height 1080 --width 1920 --batch 4 --dtype bf16 > gpurun_out/${B}_passes.log 2>&1
python tools/profile_merge.py gpurun_out/$B gpurun_out/$B > gpurun_out/${B}_merge.log 2>&1
rm -rf gpurun_out/$A/*/ gpurun_out/$B/*/ 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && R=$OLDPWD && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${A}_benchtrace -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/${A}_bench_under_rocprof.json 2> $R/gpurun_out/${A}_bench_under_rocprof.err )
cp $(find gpurun_out/${A}_benchtrace -name "*kernel_stats.csv" | head -1) gpurun_out/${A}_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/${A}_benchtrace
python bench.py > gpurun_out/${A}_bench.json 2> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --dtype bf16 --no-cpu-baseline --no-other-configs > gpurun_out/${A}_bench_bf16.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --height 1080 --width 1920 --batch 4 --dtype bf16 --no-cpu-baseline > gpurun_out/${B}_bench_1080p_bf16.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --height 1080 --width 1920 --batch 4 --no-cpu-baseline > gpurun_out/${B}_bench_1080p_f32.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --height 640 --width 1120 --no-cpu-baseline --no-other-configs > gpurun_out/${A}_bench_640x1120.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --variant lolv1 --height 400 --width 600 --no-cpu-baseline --no-other-configs > gpurun_out/${A}_bench_lolv1_400x600.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --fdsa-pair --no-cpu-baseline --no-other-configs > gpurun_out/${A}_bench_round5_routing.json 2>> gpurun_out/${A}_bench.err
head -4 gpurun_out/${A}_summary.txt; head -3 gpurun_out/${B}_summary.txt; cat gpurun_out/${A}_gputests.txt
