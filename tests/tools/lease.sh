#!/bin/bash
# One GPU lease = one call of this script on the GPU box:   gpurun -- bash tests/tools/lease.sh <name> <task> [<task> ...]
# Everything a task writes goes to gpurun_out/<name>/ (merged back; what is worth judging is copied to profiles/ by hand).
# Tasks (each under its own timeout; a failing task does not stop the following ones):
#   suite            pytest -m gpu (full GPU suite)                                   -> gpu_tests.log
#   smoke            __graft_entry__.smoke()                                          -> smoke.log
#   bench            python bench.py (default line: headline + legs + CPU baseline)   -> bench_n1.json / .err
#   bench:<args>     python bench.py <args> ("," stands for a blank)                  -> bench_<n>.json / .err
#   stats            rocprofv3 --kernel-trace --stats of the headline leg, its launches   -> kernel_stats.csv
#                    (5 M reads each) one after the other: --contexts 1 --chunk 5000000
#   stats_greedy     ... of --mode greedy                                             -> kernel_stats_greedy.csv
#   pmc              PMC passes of the search kernels (tests/tools/pmc_bench.sh) + profiles/traffic.json
#   refseq_ref       BASELINE configs[3] at its named scale (28 G rows: 14.3 M proteins x 7): bench.py --image --paired
#   refseq           BASELINE configs[4], single-GPU half (56 G rows: 14.3 M proteins x 14), the .fmi streamed to HBM, no image
#   py:<script>      python <script> ("," stands for a blank)                         -> py_<n>.log
#   sh:<script>      bash <script>                                                    -> sh_<n>.log
cd "$GRAFT_REPO_ROOT" || exit 1
NAME=$1; shift
O=gpurun_out/$NAME; mkdir -p "$O"
export TMPDIR=/tmp
n=0
for task in "$@"; do
  n=$((n + 1))
  t0=$(date +%s)
  case "$task" in
    suite)
      ( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[lease] suite rc=$?"; tail -4 $O/gpu_tests.log ;;
    smoke)
      timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "[lease] smoke rc=$?"; tail -3 $O/smoke.log ;;
    bench)
      timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[lease] bench rc=$?"; tail -6 $O/bench_n1.err ;;
    bench:*)
      args=$(echo "${task#bench:}" | tr ',' ' ')
      timeout 2400 python bench.py $args > $O/bench_$n.json 2> $O/bench_$n.err; echo "[lease] bench $args rc=$?"; tail -6 $O/bench_$n.err ;;
    stats)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 --contexts 1 --chunk 5000000 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
      cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats; echo "[lease] stats"; head -8 $O/kernel_stats.csv ;;
    stats_greedy)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_g -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode greedy --legs "" --steps 3 > $GRAFT_REPO_ROOT/$O/bench_greedy_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_greedy_under_rocprof.err )
      cp $O/stats_g/s_kernel_stats.csv $O/kernel_stats_greedy.csv 2>/dev/null; rm -rf $O/stats_g; echo "[lease] stats_greedy"; head -6 $O/kernel_stats_greedy.csv ;;
    pmc)
      bash tests/tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; tail -5 $O/pmc.log
      python tests/tools/pmc_bench_collect.py $O/pmc $O/traffic.json profiles/${PMC_RAW_NAME:-r05_pmc} > $O/pmc_collect.log 2>&1; tail -3 $O/pmc_collect.log ;;
    refseq_ref)
      # BASELINE configs[3]: refseq_ref class, 28 G rows.  The box's cgroup allows 300 GiB of memory INCLUDING /dev/shm (the host
      # has 3 TB, / only 79 GB): a real sort of 28 G suffixes (224 GB of 64-bit positions) does not fit, so the index is the one of
      # a 4.0 G-row database (14.3 M proteins, sorted for real) with every protein seven times (kaiju_build_fmi_replicated);
      # a refseq-class index of 58 G rows (configs[4]: .fmi 122 GB + image 150 GB) cannot be staged on this box at all.
      # A watchdog ends the task before the cgroup would (a box driven out of memory is a strike).
      W=/dev/shm/kaiju_big_$task; mkdir -p $W
      NSEQ=14300001; ARGS="--copies 7 --paired --reads 5000000 --steps 10 --legs greedy --leg-steps 2 --cpu-sample 200000 --cpu-sample-legs 100000"
      # the bench runs in a process group of its own, so that the watchdog can end exactly that group
      setsid bash -c "KAIJU_GPU_LOAD_TIMES=1 exec timeout ${LEASE_BIG_TIMEOUT:-1700} python bench.py --work $W --nseq $NSEQ --image --warmup 1 --no-ref-ops $ARGS > $O/bench_$task.json 2> $O/bench_$task.err" &
      BP=$!
      ( while sleep 2; do
          kill -0 $BP 2>/dev/null || break
          cur=$(cat /sys/fs/cgroup/memory.current 2>/dev/null || echo 0)
          echo "$(date +%s) $cur" >> $O/memory_$task.txt
          if [ "$cur" -gt 285000000000 ]; then echo "[lease] memory watchdog: $cur bytes - ending the task" >> $O/bench_$task.err; kill -KILL -- -$BP; rm -rf $W; break; fi
        done ) &
      WD=$!
      wait $BP
      echo "[lease] $task rc=$?"; grep -v "^\[kaiju_gpu pack\]" $O/bench_$task.err | tail -40; ls -la $W > $O/files_$task.txt; df -h /dev/shm >> $O/files_$task.txt; free -g >> $O/files_$task.txt
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$task -o s -- python $GRAFT_REPO_ROOT/bench.py --work $W --nseq $NSEQ --image --no-cpu-baseline --legs "" --steps 3 --warmup 1 $(echo $ARGS | sed 's/--steps [0-9]*//; s/--legs [a-z]*//') > $GRAFT_REPO_ROOT/$O/bench_${task}_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_${task}_under_rocprof.err )
      cp $O/stats_$task/s_kernel_stats.csv $O/kernel_stats_$task.csv 2>/dev/null; rm -rf $O/stats_$task; head -6 $O/kernel_stats_$task.csv
      kill $WD 2>/dev/null
      sort -k2 -n $O/memory_$task.txt | tail -1 > $O/memory_peak_$task.txt; rm -f $O/memory_$task.txt
      rm -rf $W ;;
    refseq)
      # BASELINE configs[4], its single-GPU half: a refseq-class index (README.md:102: 111 GB .fmi) on ONE MI355X.  The 4.0 G-row
      # database of refseq_ref above with every protein FOURTEEN times = 56 G rows, .fmi ~107 GB in /dev/shm - and nothing else
      # staged: the .fmi is streamed to HBM and packed there (fmi_stream.hip), no image, no host copy.  The .fmi is built by a
      # process of its own (--prepare-only) so that the bench process's peak resident set is the LOADER's (+ the reads).
      # Every step runs in a process group of its own; a watchdog ends exactly that group before the cgroup (300 GiB incl.
      # /dev/shm) would end the box.
      W=/dev/shm/kaiju_big_$task; mkdir -p $W
      NSEQ=14300001; COPIES=${LEASE_REFSEQ_COPIES:-14}
      ARGS="--copies $COPIES --paired --reads 5000000 --steps 10 --legs greedy --leg-steps 2 --cpu-sample 100000 --cpu-sample-legs 100000"
      guarded() {        # guarded <log of stdout> <log of stderr> <command...>
        local out=$1 err=$2; shift 2
        setsid bash -c "exec $*" > $out 2> $err &
        local bp=$!
        ( while sleep 2; do
            kill -0 $bp 2>/dev/null || break
            cur=$(cat /sys/fs/cgroup/memory.current 2>/dev/null || echo 0)
            echo "$(date +%s) $cur" >> $O/memory_$task.txt
            if [ "$cur" -gt 300000000000 ]; then echo "[lease] memory watchdog: $cur bytes - ending the step" >> $err; kill -KILL -- -$bp; rm -rf $W; break; fi
          done ) &
        local wd=$!
        wait $bp; local rc=$?
        kill $wd 2>/dev/null
        return $rc
      }
      t1=$(date +%s)
      guarded $O/prepare_$task.log $O/prepare_$task.err timeout 1500 python bench.py --work $W --nseq $NSEQ --copies $COPIES --prepare-only
      echo "[lease] prepare rc=$? ($(( $(date +%s) - t1 )) s)"; grep -v "kaiju mkfmi\]" $O/prepare_$task.err | tail -5
      ls -la $W > $O/files_$task.txt; df -h /dev/shm >> $O/files_$task.txt; free -g >> $O/files_$task.txt
      KAIJU_GPU_LOAD_TIMES=1 guarded $O/bench_$task.json $O/bench_$task.err timeout ${LEASE_BIG_TIMEOUT:-1500} python bench.py --work $W --nseq $NSEQ --warmup 1 --no-ref-ops $ARGS
      echo "[lease] $task rc=$?"; grep -v "^\[kaiju_gpu pack\]" $O/bench_$task.err | tail -40
      if [ -d $W ]; then
        ( cd /tmp && KAIJU_GPU_LOAD_TIMES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$task -o s -- python $GRAFT_REPO_ROOT/bench.py --work $W --nseq $NSEQ --no-cpu-baseline --legs "" --steps 3 --warmup 1 --copies $COPIES --paired --reads 5000000 > $GRAFT_REPO_ROOT/$O/bench_${task}_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_${task}_under_rocprof.err )
        cp $O/stats_$task/s_kernel_stats.csv $O/kernel_stats_$task.csv 2>/dev/null; rm -rf $O/stats_$task; head -8 $O/kernel_stats_$task.csv
      fi
      sort -k2 -n $O/memory_$task.txt | tail -1 > $O/memory_peak_$task.txt; rm -f $O/memory_$task.txt
      rm -rf $W ;;
    py:*)
      args=$(echo "${task#py:}" | tr ',' ' ')
      timeout 2400 python $args > $O/py_$n.log 2>&1; echo "[lease] python $args rc=$?"; tail -15 $O/py_$n.log ;;
    sh:*)
      timeout 2400 bash ${task#sh:} $O > $O/sh_$n.log 2>&1; echo "[lease] ${task#sh:} rc=$?"; tail -15 $O/sh_$n.log ;;
    *) echo "[lease] unknown task $task" ;;
  esac
  echo "[lease] $task took $(( $(date +%s) - t0 )) s"
done
du -sh $O
