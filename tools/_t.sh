cd /root/repo
tools/fuzz_hunt.sh 519 534
