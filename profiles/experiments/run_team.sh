for n in 1000000000 40000 20000 8000; do echo "team_min $n"; MM2GB_RMQ_TEAM_MIN_ANCHORS=$n timeout 600 python profiles/experiments/e2e_knobs.py 4 2x96,4x96 2>&1 | tail -2 | cut -c1-300; done
