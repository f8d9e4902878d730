// TEST INFRASTRUCTURE — not part of the product path.
//
// A small driver of OUR OWN around the reference's unmodified loaders (ScafDpData, ClusterData): reads a .lrn
// with the reference's ClusterData::ClusterData(esom_lrn_file, scaf_db) (reference src/ClusterData.cpp:27-168) and
// prints, for every datapoint NAME in [1, max_name] that the table holds, its internal id and its value in every
// dimension as %.17g.  tests/golden/make_golden.py turns the output into the golden vectors of abw_lrn_parse.
//
// With two more arguments it also writes ONE cluster the reference's way: the subset of the table whose datapoint NAMES
// are multiples of <every> goes through ClusterData(parent, subset) and the reference's unmodified ClusterEsomWriter
// (reference src/ClusterWriter.cpp:8-19,69-100) into <out_dir>/c1.lrn -- the golden vector of abw_lrn_emit_rows.
//
// usage: lrn_dump <names> <fasta> <info> <lrn> <max_name> [<out_dir> <every>]
#include <stdio.h>
#include <stdlib.h>
#include "ScafDpData.h"
#include "ClusterData.h"
#include "ClusterWriter.h"

int main(int argc, const char** argv)
{
	if(argc < 6) { fprintf(stderr, "usage: %s <names> <fasta> <info> <lrn> <max_name>\n", argv[0]); return 2; }
	ScafDpData	scaf_db(argv[1], argv[2], argv[3]);
	ClusterData	db(argv[4], scaf_db);
	const size_t max_name = strtoul(argv[5], NULL, 10);
	printf("%lu\t%lu\n", (unsigned long)db.ndps(), (unsigned long)db.ndimensions());
	for(size_t name = 1; name <= max_name; name++) {
		const size_t iid = scaf_db.dp_name2id(name);
		if(iid == 0 || db.datapoints().find(iid) == db.datapoints().end()) continue;
		printf("%lu\t%lu", (unsigned long)name, (unsigned long)iid);
		for(size_t d = 1; d <= db.ndimensions(); d++) printf("\t%.17g", db.get_dimension(d).get_value(iid));
		printf("\n");
	}
	if(argc >= 8) {
		const size_t every = strtoul(argv[7], NULL, 10);
		unordered_set<size_t> subset;
		set<size_t> scafs;
		for(size_t name = 1; name <= max_name; name++) {
			const size_t iid = scaf_db.dp_name2id(name);
			if(iid == 0 || db.datapoints().find(iid) == db.datapoints().end() || (name % every) != 0) continue;
			subset.insert(iid);
			scafs.insert(scaf_db.dp2scaf(iid));
		}
		ClusterData		node(db, subset);
		ClusterEsomWriter	writer(argv[6], scaf_db);
		writer.write(node, scafs, subset, "c1");
	}
	return 0;
}
