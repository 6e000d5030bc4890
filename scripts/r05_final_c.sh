#!/bin/bash
# round 5, final records on the final build, one box: the classic profile (so that the bench line's roofline reads the counters of
# THIS build), then the bench lines / rehearsal / config 5 of r05_final_b.sh
set -e
cd "$(dirname "$0")/.."
scripts/r05_profile.sh classic
echo "profile done" >> gpurun_out/heartbeat.txt
scripts/r05_final_b.sh
