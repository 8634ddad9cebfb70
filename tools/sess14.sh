O=gpurun_out
bash tools/profile_session.sh r3 2>&1 | tail -40
bash tools/profile_session.sh r3_stats --stats 2>&1 | tail -12
bash tools/profile_session.sh r3_dense --dense 2>&1 | tail -12
