// See ClusterSeparatorHipSharded.h.
#include "ClusterSeparatorHipSharded.h"
#include "abawaca_hip.h"
#include <stdlib.h>
#include <algorithm>
#include <stdexcept>
#include <string>

namespace {

struct Shard {
	abw_ctx*		ctx = NULL;
	int32_t			first_scaf = 0, n_scaf = 0;	// global scaffold ids [first_scaf, first_scaf + n_scaf)
	int64_t			first_dp = 0, n_dp = 0;
	std::vector<int32_t>	node_ids;			// the current node's scaffolds on this shard, local ids
	std::vector<uint8_t>	side, raw;
};

struct ShardedStore {
	std::vector<Shard>	shards;
	const void*		owner = NULL;			// the ScafDpData the table was built from
	std::vector<int32_t>	shard_of;			// [S] shard of every scaffold
	std::vector<int32_t>	scaf_ndps;			// [S]
};
ShardedStore g_sharded;

void fail(const std::string& what, abw_ctx* ctx)
{
	throw std::runtime_error(ctx ? what + ": " + abw_last_error(ctx) : what);
}

}	// namespace

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHipSharded::shutdown()
{
	for(size_t k = 0; k < g_sharded.shards.size(); k++)
		if(g_sharded.shards[k].ctx) abw_ctx_destroy(g_sharded.shards[k].ctx);
	g_sharded = ShardedStore();
	ClusterSeparatorHip::shutdown();
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
size_t ClusterSeparatorHipSharded::nshards()
{
	return g_sharded.shards.size();
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorHipSharded::loaded_for(const ScafDpData* db) const
{
	return !g_sharded.shards.empty() && g_sharded.owner == (const void*)db;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHipSharded::load_table(const Table& t)
{
	ShardedStore& G = g_sharded;
	for(size_t k = 0; k < G.shards.size(); k++)
		if(G.shards[k].ctx) abw_ctx_destroy(G.shards[k].ctx);
	G = ShardedStore();
	G.owner = (const void*)t.owner;

	int32_t ndev = 0;
	if(abw_device_count(&ndev) != ABW_OK || ndev <= 0)
		throw std::runtime_error("ClusterSeparatorHipSharded: no MI355X / HIP device available");
	const char* env = getenv("ABW_SHARDS");
	long want = env ? strtol(env, NULL, 10) : ndev;
	if(want < 1 || want > 64) throw std::invalid_argument("ClusterSeparatorHipSharded: ABW_SHARDS must be in [1,64]");
	const size_t K = std::min((size_t)want, t.S);

	// scaffolds in contiguous ranges of about N / K datapoints each
	G.scaf_ndps.assign(t.S, 0);
	for(size_t dp = 0; dp < t.N; dp++) G.scaf_ndps[t.dp2scaf[dp]]++;
	G.shard_of.assign(t.S, 0);
	G.shards.assign(K, Shard());
	{
		size_t k = 0;
		int64_t seen = 0;
		G.shards[0].first_scaf = 0; G.shards[0].first_dp = 0;
		for(size_t s = 0; s < t.S; s++) {
			// the next shard starts where the datapoints seen reach its share (and every shard gets a scaffold)
			while(k + 1 < K && G.shards[k].n_scaf > 0 && (seen * (int64_t)K >= (int64_t)(k + 1) * (int64_t)t.N || t.S - s <= K - 1 - k)) {
				k++;
				G.shards[k].first_scaf = (int32_t)s; G.shards[k].first_dp = seen;
			}
			G.shard_of[s] = (int32_t)k;
			G.shards[k].n_scaf++;
			G.shards[k].n_dp += G.scaf_ndps[s];
			seen += G.scaf_ndps[s];
		}
	}
	// one bin layout for all shards (abw_sep_shard_load): a column whose keys all lie below 1024 keeps the key as bin
	// (empty dictionary); any other column's dictionary is its sorted distinct keys over the WHOLE table -- the keys are
	// ranks among the column's distinct values (ClusterSeparatorHip::Table), so that is 0 .. max
	std::vector<int64_t> dict_off(t.D + 1, 0);
	std::vector<int32_t> dict_vals;
	for(size_t d = 0; d < t.D; d++) {
		int32_t mx = 0;
		for(size_t p = 0; p < t.N; p++) mx = std::max(mx, t.keys[d * t.N + p]);
		if(mx >= 1024) for(int32_t v = 0; v <= mx; v++) dict_vals.push_back(v);
		dict_off[d + 1] = (int64_t)dict_vals.size();
	}
	for(size_t k = 0; k < K; k++) {
		Shard& sh = G.shards[k];
		if(abw_ctx_create((int)(k % (size_t)ndev), &sh.ctx) != ABW_OK)
			throw std::runtime_error("ClusterSeparatorHipSharded: cannot create a context on device " + std::to_string(k % (size_t)ndev));
		std::vector<int32_t> keys((size_t)t.D * sh.n_dp), dp2scaf(sh.n_dp), scg_off(sh.n_scaf + 1, 0), scg_ids;
		for(size_t d = 0; d < t.D; d++)
			std::copy(t.keys.begin() + d * t.N + sh.first_dp, t.keys.begin() + d * t.N + sh.first_dp + sh.n_dp, keys.begin() + d * sh.n_dp);
		for(int64_t p = 0; p < sh.n_dp; p++) dp2scaf[p] = t.dp2scaf[sh.first_dp + p] - sh.first_scaf;
		if(t.n_scg) {
			scg_ids.assign(t.scg_ids.begin() + t.scg_off[sh.first_scaf], t.scg_ids.begin() + t.scg_off[sh.first_scaf + sh.n_scaf]);
			for(int32_t s = 0; s <= sh.n_scaf; s++) scg_off[s] = t.scg_off[sh.first_scaf + s] - t.scg_off[sh.first_scaf];
		}
		if(abw_sep_shard_load(sh.ctx, keys.data(), (int32_t)t.D, sh.n_dp, dp2scaf.data(), t.scaf_len.data() + sh.first_scaf, sh.n_scaf,
				t.n_scg ? scg_off.data() : NULL, t.n_scg ? scg_ids.data() : NULL, t.n_scg, dict_vals.empty() ? NULL : dict_vals.data(),
				dict_off.data()) != ABW_OK)
			fail("ClusterSeparatorHipSharded: abw_sep_shard_load failed on shard " + std::to_string(k), sh.ctx);
	}
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHipSharded::run_node(const std::vector<int32_t>& scaf_ids, size_t n_dps, NodeResult& r)
{
	ShardedStore& G = g_sharded;
	const size_t K = G.shards.size();
	const int32_t min_ndps = (int32_t)this->cluster_ndps_threshold;
	for(size_t k = 0; k < K; k++) G.shards[k].node_ids.clear();
	for(size_t i = 0; i < scaf_ids.size(); i++) {
		Shard& sh = G.shards[G.shard_of[scaf_ids[i]]];
		sh.node_ids.push_back(scaf_ids[i] - sh.first_scaf);
	}
	// accumulate everywhere (asynchronous), combine, evaluate everywhere
	std::vector<abw_ctx*> ctxs(K);
	for(size_t k = 0; k < K; k++) {
		Shard& sh = G.shards[k];
		abw_sep_exchange ex;
		ctxs[k] = sh.ctx;
		if(abw_sep_shard_accum(sh.ctx, sh.node_ids.empty() ? NULL : sh.node_ids.data(), (int32_t)sh.node_ids.size(), &ex) != ABW_OK)
			fail("ClusterSeparatorHipSharded: abw_sep_shard_accum failed", sh.ctx);
	}
	if(abw_sep_shard_combine(ctxs.data(), (int32_t)K) != ABW_OK) {
		for(size_t k = 0; k < K; k++) if(abw_last_error(ctxs[k])[0]) fail("ClusterSeparatorHipSharded: abw_sep_shard_combine failed", ctxs[k]);
		fail("ClusterSeparatorHipSharded: abw_sep_shard_combine failed", NULL);
	}
	abw_split out, first;
	int64_t n1 = 0, n2 = 0;
	for(size_t k = 0; k < K; k++) {
		Shard& sh = G.shards[k];
		size_t nd = 0;
		for(size_t i = 0; i < sh.node_ids.size(); i++) nd += G.scaf_ndps[sh.first_scaf + sh.node_ids[i]];
		sh.side.assign(sh.node_ids.size() + 1, 0);
		sh.raw.assign(nd + 1, 0);
		if(abw_sep_shard_eval(sh.ctx, min_ndps, &out, NULL, sh.side.data(), sh.raw.data()) != ABW_OK)
			fail("ClusterSeparatorHipSharded: abw_sep_shard_eval failed", sh.ctx);
		if(k == 0) first = out;
		else if(out.dimension != first.dimension || out.key != first.key || out.legal != first.legal ||
			out.sensitivity != first.sensitivity || out.specificity != first.specificity)
			throw std::runtime_error("ClusterSeparatorHipSharded: the shards disagree on a node's winner");
		n1 += out.ndps1; n2 += out.ndps2;
	}
	out = first;
	abw_sep_shard_decide(&out, n1, n2, min_ndps);				// reference src/ClusterSeparator.cpp:125-132
	r.dimension = out.dimension; r.key = out.key; r.legal = out.legal; r.split = out.split;
	r.sensitivity = out.sensitivity; r.specificity = out.specificity;
	// back into the order of scaf_ids (ascending global id = shard by shard, local order inside)
	r.side.assign(scaf_ids.size(), 0);
	r.raw.assign(n_dps, 0);
	std::vector<size_t> at_s(K, 0), at_d(K, 0);
	size_t d = 0;
	for(size_t i = 0; i < scaf_ids.size(); i++) {
		const size_t k = (size_t)G.shard_of[scaf_ids[i]];
		Shard& sh = G.shards[k];
		r.side[i] = (first.legal && !out.split) ? 0 : sh.side[at_s[k]];	// as abw_sep_node: no sides behind a reset
		at_s[k]++;
		const size_t n = (size_t)G.scaf_ndps[scaf_ids[i]];
		for(size_t j = 0; j < n; j++) r.raw[d++] = sh.raw[at_d[k]++];
	}
	// the raw sides stay (the reference keeps its raw clusters across the reset, reference src/ClusterSeparator.cpp:87-90,125-132)
	r.legal = first.legal;
}
