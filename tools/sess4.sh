bash tools/gpu_session_r3.sh tests
bash tools/kb_session.sh "classic_nohand 0 7 -" "classicA_nohand 0 7 - -DADSB_ABLATE=2" "pipeNoB_nohand 1 5 - -DADSB_PIPE_ABLATE=1" "pipe_nohand 1 5 -" "pipeNoB 1 5 - -DADSB_PIPE_ABLATE=1" "classic 0 7 -"
