// ClusterSeparatorHipSharded — ClusterSeparatorHip with the table spread over several GPU contexts of THIS process
// ("separator sweep" row; the reference's driver is one process, reference src/abawaca.cpp:95-197).
//
// The scaffolds are cut into ABW_SHARDS contiguous ranges of about equal numbers of datapoints (environment; default: one
// per visible device); shard k lives in its own abw_ctx on device k mod (visible devices), loaded with abw_sep_shard_load
// under one bin layout.  Per node every shard accumulates the node's scaffolds it holds (abw_sep_shard_accum), the
// histograms and SCG lists are combined over the contexts (abw_sep_shard_combine: peer copies and an add kernel), every
// shard finds the same winner and assigns its own scaffolds (abw_sep_shard_eval), and abw_sep_shard_decide applies the
// size check on the summed children.  All exchanged quantities are integers: the result is the one of ClusterSeparatorHip
// bit for bit, for any number of shards.  Same constructor arguments, same swap at reference src/abawaca.cpp:109.
#ifndef CLUSTERSEPARATORHIPSHARDED_H
#define CLUSTERSEPARATORHIPSHARDED_H

#include "ClusterSeparatorHip.h"

class ClusterSeparatorHipSharded : public ClusterSeparatorHip {
public:
	ClusterSeparatorHipSharded(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore) :
		ClusterSeparatorHip(scaf_db, scg_db, clustering_db, threads_semaphore)	{}
	// Releases the contexts and their shards (optional; call once at program end).
	static void	shutdown();
	// number of shards in use (0 before the first separate())
	static size_t	nshards();
protected:
	void		load_table(const Table& t) override;
	void		run_node(const std::vector<int32_t>& scaf_ids, size_t n_dps, NodeResult& r) override;
	bool		loaded_for(const ScafDpData* db) const override;
	bool		run_tree(std::vector<TreeRec>&, std::vector<int32_t>&) override	{return false;}	// shards go node by node (abw_sep_shard_*)
};

#endif // CLUSTERSEPARATORHIPSHARDED_H
