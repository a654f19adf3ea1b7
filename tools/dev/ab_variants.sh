for n in A B C D E F A; do echo "== variant $n"; GDPT_LIB=$PWD/gradientdomain-mitsuba_amd/lib/var/libgdpt_$n.so timeout 300 python tools/gpu_quick_perf.py 2>&1 | tail -3; done
