for seg in 64 128 256 1024; do for mode in 2 4 0 5; do ./tools/membench 384 1038240 $seg $mode 0; done; done
