for seg in 128 256 512 1024; do ./tools/membench 384 1038240 $seg 3 0; ./tools/membench 384 1038240 $seg 1 0; done
./tools/membench 384 1038240 128 3 0 512
