cd $GRAFT_REPO_ROOT
./tests/cpp/test_dropin tests/golden/words.txt 2>&1 | tail -12
