// TEST INFRASTRUCTURE — not part of the product path.
//
// A small driver of OUR OWN that links the reference's unmodified objects (ScafDpData,
// ClusterData, SCGdb, ClusterSeparatorBySensitivitySpecificity, ...) and runs the same
// work-list recursion as the reference's main loop (reference src/abawaca.cpp:95-197), but
//   * takes the SCG-list path as an argument (the reference hard-codes an absolute path,
//     src/abawaca.cpp:42, which does not exist here),
//   * skips the fasta output, and writes the clusters' files (ClusterEsomWriter, reference src/abawaca.cpp:161-182)
//     only when the environment names a directory for them: ABW_CLUSTER_DIR=<dir>; the GPU builds then use
//     integration/ClusterEsomWriterHip instead.  <out_prefix>.writer.tsv: microseconds spent writing, clusters written,
//   * writes machine-readable results with full double precision, and the wall time of every
//     separate() call (this is what bench.py's cpu_baseline "reference" leg reads).
//
// usage: ref_sep_driver <names> <fasta> <info> <lrn> <gene2scg|-> <scg.list> <ncpus> <out_prefix> [max_nodes]
//
// outputs:
//   <out_prefix>.nodes.tsv        one row per visited node:
//        id ndps split dim value specificity sensitivity child1 ndps1 child2 ndps2 usec
//   <out_prefix>.scaf2cluster.tsv scaffold-name <tab> terminal cluster id (0 = none), ids 1..S
//   <out_prefix>.dp2cluster.tsv   internal dp id <tab> terminal cluster id
//   <out_prefix>.dpmap.tsv        internal dp id <tab> internal scaffold id   (the ScafDpData numbering)
//   <out_prefix>.summary.txt      the reference's summary.txt: its header and one row per terminal cluster, computed by the
//        reference's unmodified ClusterQuality / SCGdb and printed with the expressions of reference src/abawaca.cpp:93,144-152
//        (scaffolds, datapoints, bps, GC, coverage, unique SCGs = completeness, average copies = contamination)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <map>
#include <string>
#include "ScafDpData.h"
#include "ClusterData.h"
#include "ClusterSeparatorBySensitivitySpecificity.h"
#if defined(ABW_USE_HIP_SHARDED_SEPARATOR)	// ... with the table spread over ABW_SHARDS contexts (integration/ClusterSeparatorHipSharded.h)
#include "ClusterSeparatorHipSharded.h"
#define SEPARATOR_CLASS ClusterSeparatorHipSharded
#define ABW_USE_HIP_SEPARATOR
#elif defined(ABW_USE_HIP_SEPARATOR)	// same driver, separator swapped for the GPU drop-in (integration/ClusterSeparatorHip.h)
#include "ClusterSeparatorHip.h"
#define SEPARATOR_CLASS ClusterSeparatorHip
#elif defined(ABW_USE_KMEANS_SEPARATOR)	// ... or for the k-means extension (integration/ClusterSeparatorKMeans.h)
#include "ClusterSeparatorKMeans.h"
#define SEPARATOR_CLASS ClusterSeparatorKMeans
#else
#define SEPARATOR_CLASS ClusterSeparatorBySensitivitySpecificity
#endif
#include "ClusterWriter.h"
#if defined(ABW_USE_HIP_SEPARATOR) || defined(ABW_USE_KMEANS_SEPARATOR)
#include "ClusterDataView.h"
#include "ClusterEsomWriterHip.h"
#endif
#include "Semaphore.h"
#include "SCGdb.h"
#include "ClusterQuality.h"
#include <fstream>

int main(int argc, const char** argv)
{
	if(argc < 9) {
		fprintf(stderr, "usage: %s <names> <fasta> <info> <lrn> <gene2scg|-> <scg.list> <ncpus> <out_prefix> [max_nodes]\n", argv[0]);
		return 2;
	}
	std::string names = argv[1], fasta = argv[2], info = argv[3], lrn = argv[4];
	std::string gene2scg = strcmp(argv[5], "-") ? argv[5] : "";
	std::string scg_list = argv[6];
	int ncpus = atoi(argv[7]);
	std::string prefix = argv[8];
	size_t max_nodes = (argc > 9) ? strtoul(argv[9], NULL, 10) : (size_t)-1;

	ScafDpData	scaf_db(names, fasta, info);
	ClusterData	clustering_db(lrn, scaf_db);
	SCGdb		scg_db(scaf_db, gene2scg, scg_list);
	Semaphore	sem(ncpus);

	FILE* fn = fopen((prefix + ".nodes.tsv").c_str(), "w");
	FILE* fm = fopen((prefix + ".dpmap.tsv").c_str(), "w");
	if(!fn || !fm) { perror("open output"); return 1; }
	for(size_t dp = 1; dp <= scaf_db.ndps(); dp++)
		fprintf(fm, "%lu\t%lu\n", dp, scaf_db.dp2scaf(dp));
	fclose(fm);

	std::ofstream summary((prefix + ".summary.txt").c_str());
	summary << "Cluster\t# scafs\t# dps\t# bps\t%G+C\tStdev\tCvg\tstdev\t#SCG" << '/' << scg_db.get_total_num_scgs() << "\tAvg" << endl;	// reference :93
	const char* cluster_dir = getenv("ABW_CLUSTER_DIR");
	long write_usec = 0, written = 0;
	std::map<size_t, Cluster*> work;
	std::map<size_t, size_t> dp2cluster, scaf2cluster;
	size_t next_id = 1, visited = 0;

	Cluster* root = new Cluster(scaf_db);
	for(size_t i = 1; i <= scaf_db.ndps(); i++) *root += i;
	for(size_t i = 1; i <= scaf_db.nscafs(); i++) root->add_assigned_scaf(i);
	work[next_id++] = root;

	while(!work.empty() && visited < max_nodes) {
		auto it = work.begin();
		size_t id = it->first;
		Cluster* cur = it->second;
		work.erase(it);
		visited++;

#if defined(ABW_USE_HIP_SEPARATOR) || defined(ABW_USE_KMEANS_SEPARATOR)
		// the GPU separators keep the table resident: after the first node a node needs only its datapoint set
		// (integration/ClusterDataView.h; this driver has no ClusterEsomWriter that would read the node's values)
		// (the root IS the loaded table: no copy of it either, unless the reference's per-node subset tables are asked for)
		const bool full_tables = getenv("ABW_FULL_NODE_TABLES") != NULL;
		ClusterData* node_db = full_tables ? new ClusterData(clustering_db, cur->get_dps())
		                     : visited == 1 ? &clustering_db : new ClusterDataView(clustering_db, cur->get_dps());
#else
		ClusterData* node_db = new ClusterData(clustering_db, cur->get_dps());
#endif
		SEPARATOR_CLASS sep(scaf_db, scg_db, *node_db, sem);

		auto t0 = std::chrono::steady_clock::now();
		bool split = sep.separate();
		auto t1 = std::chrono::steady_clock::now();
		long usec = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();

		if(!split || sep.get_cluster1() == NULL) {
			fprintf(fn, "%lu\t%lu\t0\t%ld\t%.17g\t%.17g\t%.17g\t0\t0\t0\t0\t%ld\n", id, cur->ndps(),
				(long)sep.get_separating_dimension(), sep.get_separating_value(),
				sep.get_best_specificity(), sep.get_best_sensitivity(), usec);
			for(auto s = cur->dps_begin(); s != cur->dps_end(); s++) dp2cluster[*s] = id;
			for(auto s = cur->assigned_scafs_begin(); s != cur->assigned_scafs_end(); s++) scaf2cluster[*s] = id;
			// the bin's row of summary.txt (reference src/abawaca.cpp:137-152)
			ClusterQuality cq(scaf_db, scg_db);
			double nunique_scgs, avg_ncopies_scgs, avg_gc, stdev_gc, avg_cvg = -1, stdev_cvg = -1;
			const size_t total_bps = cq.total_size(*cur), nscaffolds = cq.num_scaffolds(*cur);
			cq.scg(*cur, nunique_scgs, avg_ncopies_scgs);
			cq.gc(*cur, avg_gc, stdev_gc);
			cq.cvg(*cur, avg_cvg, stdev_cvg);
			summary << id << '\t' << nscaffolds << '\t' << cur->ndps() << '\t' << total_bps << '\t'
				<< int(1000*avg_gc)/10.0 << '\t' << int(1000*stdev_gc)/100.0 << '\t' << int(10*avg_cvg)/10.0
				<< '\t' << int(10*stdev_cvg)/10.0 << '\t' << nunique_scgs << '\t' << int(100*avg_ncopies_scgs)/100.0 << endl;
		}
		else {
			size_t c1 = next_id++, c2 = next_id++;
			fprintf(fn, "%lu\t%lu\t1\t%ld\t%.17g\t%.17g\t%.17g\t%lu\t%lu\t%lu\t%lu\t%ld\n", id, cur->ndps(),
				(long)sep.get_separating_dimension(), sep.get_separating_value(),
				sep.get_best_specificity(), sep.get_best_sensitivity(),
				c1, sep.get_cluster1()->ndps(), c2, sep.get_cluster2()->ndps(), usec);
			if(cluster_dir) {		// reference src/abawaca.cpp:161-182
				auto w0 = std::chrono::steady_clock::now();
				char name1[128], name2[128];
				sprintf(name1, "%lu", c1);
				sprintf(name2, "%lu", c2);
#if defined(ABW_USE_HIP_SEPARATOR) || defined(ABW_USE_KMEANS_SEPARATOR)
				ClusterEsomWriterHip writer(cluster_dir, scaf_db, clustering_db);
				ClusterDataView cluster1(clustering_db, sep.get_cluster1()->get_dps()), cluster2(clustering_db, sep.get_cluster2()->get_dps());
#else
				ClusterEsomWriter writer(cluster_dir, scaf_db);
				ClusterData cluster1(*node_db, sep.get_cluster1()->get_dps()), cluster2(*node_db, sep.get_cluster2()->get_dps());
#endif
				writer.write(cluster1, sep.get_cluster1()->get_assigned_scafs(), sep.get_raw_dps_cluster1(), name1);
				writer.write(cluster2, sep.get_cluster2()->get_assigned_scafs(), sep.get_raw_dps_cluster2(), name2);
				write_usec += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - w0).count();
				written += 2;
			}
			work[c1] = sep.get_cluster1();
			work[c2] = sep.get_cluster2();
		}
		fflush(fn);
		if(node_db != &clustering_db) delete node_db;
		delete cur;
	}
	fclose(fn);
	if(cluster_dir) {
		FILE* fw = fopen((prefix + ".writer.tsv").c_str(), "w");
		if(fw) { fprintf(fw, "%ld\t%ld\n", write_usec, written); fclose(fw); }
	}

#if defined(ABW_USE_HIP_SEPARATOR) && !defined(ABW_USE_HIP_SHARDED_SEPARATOR)
	{	// how the adapter answered: from the one-call recursion's records / node by node
		size_t from_tree = 0, node_by_node = 0;
		ClusterSeparatorHip::stats(from_tree, node_by_node);
		FILE* fx = fopen((prefix + ".sep_stats.tsv").c_str(), "w");
		if(fx) { fprintf(fx, "%lu\t%lu\n", from_tree, node_by_node); fclose(fx); }
	}
#endif
	FILE* fs = fopen((prefix + ".scaf2cluster.tsv").c_str(), "w");
	FILE* fd = fopen((prefix + ".dp2cluster.tsv").c_str(), "w");
	if(!fs || !fd) { perror("open output"); return 1; }
	for(size_t i = 1; i <= scaf_db.nscafs(); i++)
		fprintf(fs, "%s\t%lu\n", scaf_db.scaf_id2name(i).c_str(), scaf2cluster.count(i) ? scaf2cluster[i] : (size_t)0);
	for(size_t i = 1; i <= scaf_db.ndps(); i++)
		fprintf(fd, "%lu\t%lu\n", i, dp2cluster.count(i) ? dp2cluster[i] : (size_t)0);
	fclose(fs);
	fclose(fd);
	return 0;
}
