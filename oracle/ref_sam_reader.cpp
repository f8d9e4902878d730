// TEST INFRASTRUCTURE — the reference's own SAM reader timed on the host (bench.py's cpu_baseline of the SAM leg) and used
// as a second check of abw_sam_parse.  Our main around Bio::SAMReader / Bio::ReadMapping (reference
// src/ReadMappingReader.cpp:80-116, src/ReadMapping.cpp:23-185), compiled from the reference's unmodified sources by
// oracle/Makefile into oracle/_ref/ref_sam_reader; the loop is the one of reference src/abawaca-build.cpp:525-552 without
// the scaffolds.  Prints: records, records kept by the filter of :546, sums of FLAG & 0x104 / ref_pos / read_length / num_snps over
// all records, seconds.
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "ReadMappingReader.h"

int main(int argc, const char* argv[])
{
	if(argc < 2) { fprintf(stderr, "usage: %s <sam-file> [max_snps]\n", argv[0]); return 2; }
	const size_t max_snps = argc > 2 ? (size_t)atol(argv[2]) : 15;
	const auto t0 = std::chrono::steady_clock::now();
	Bio::SAMReader reader(argv[1]);
	unsigned long long n = 0, kept = 0, s_flag = 0, s_pos = 0, s_len = 0, s_snps = 0;
	while(reader.good()) {
		Bio::ReadMappingPtr mapping(reader.next_mapping());
		if(mapping == NULL)
			break;
		n++;
		s_flag += (mapping->unmapped() ? 4 : 0) + (mapping->multiple_hits() ? 256 : 0); s_pos += mapping->ref_pos(); s_len += mapping->read_length(); s_snps += mapping->num_snps();
		if(mapping->unmapped() || (mapping->num_snps() > max_snps) || mapping->multiple_hits())
			continue;
		kept++;
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%llu %llu %llu %llu %llu %llu %.6f\n", n, kept, s_flag, s_pos, s_len, s_snps, dt);
	return 0;
}
