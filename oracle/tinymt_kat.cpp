// TEST INFRASTRUCTURE — not part of the product path.
// Prints the first <n> tinymt32_generate_uint32 outputs of the reference's vendored TinyMT32
// (reference src/TinyMT-src-1.0.1/tinymt/tinymt32.{c,h}) for the parameter triple that
// reference src/misc.cpp:29-31 uses, for each seed given on the command line.
// Used by tests/golden/make_golden.py to extend the upstream KAT (check32.out.txt, seed 1)
// to further seeds.   usage: tinymt_kat <n> <seed> [<seed> ...]
#include <stdio.h>
#include <stdlib.h>
#include "tinymt32.h"

int main(int argc, char** argv)
{
	if(argc < 3) return 2;
	int n = atoi(argv[1]);
	for(int a = 2; a < argc; a++) {
		tinymt32_t t;
		t.mat1 = 0x8f7011eeU; t.mat2 = 0xfc78ff1fU; t.tmat = 0x3793fdffU;
		uint32_t seed = (uint32_t)strtoul(argv[a], NULL, 10);
		tinymt32_init(&t, seed);
		printf("%u", seed);
		for(int i = 0; i < n; i++) printf(" %u", tinymt32_generate_uint32(&t));
		printf("\n");
	}
	return 0;
}
