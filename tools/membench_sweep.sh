for seg in 64 128 256 512 1024 4096; do for mode in 0 1 2; do ./tools/membench 384 1038240 $seg $mode 0; done; done
./tools/membench 384 1038240 128 0 1
./tools/membench 384 1038240 256 0 1
./tools/membench 384 1038240 128 0 0 512
./tools/membench 384 1038240 128 0 0 1024
./tools/membench 384 1038240 256 0 0 1024
