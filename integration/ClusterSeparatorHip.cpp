// See ClusterSeparatorHip.h.  Host-side adapter between the reference's data model (ScafDpData,
// ClusterData, SCGdb, Cluster) and the C ABI of libabawaca_hip.so (include/abawaca_hip.h).
#include "ClusterSeparatorHip.h"
#include "abawaca_hip.h"
#include <stdlib.h>
// the GPU this process works on: ABW_DEVICE in the environment (default 0)
static inline int abw_env_device() { const char* e = getenv("ABW_DEVICE"); return e ? atoi(e) : 0; }
#include <algorithm>
#include <map>
#include <thread>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

// One resident column store per process: the root ClusterData is uploaded the first time a
// separator is asked to work (reference src/abawaca.cpp:86 loads it once, too).
struct HipStore {
	abw_ctx*				ctx = NULL;
	const ScafDpData*			scaf_db = NULL;
	size_t					N = 0, D = 0, S = 0;
	std::vector<std::vector<double> >	values;		// [D] sorted distinct values: key -> .lrn value
	std::vector<std::string>		dim_names;
	std::vector<uint8_t>			raw_of;		// [N + 1] scratch of separate(): raw side of a datapoint
	std::vector<size_t>			first_dp;	// [S + 1] smallest datapoint id of every scaffold (its datapoints are consecutive ids)
	std::vector<uint32_t>			cnt;		// [S + 1] scratch of separate(): datapoints of the node per scaffold (all zero between calls)
	// the whole recursion, run at the root (ClusterSeparatorHip::run_tree)
	std::vector<ClusterSeparatorHip::TreeRec>	tree;	// in id order = the order the reference's driver visits them
	std::vector<int32_t>			s2c;		// [S] final bin (terminal node id) of every scaffold
	std::vector<int32_t>			idx_of, tin, tout;	// node id -> index in tree; depth-first intervals by index
	size_t					cursor = 0;	// the next separate() call is tree[cursor]
	size_t					from_tree = 0, node_by_node = 0;
};
HipStore g_store;

void fail(const std::string& what)
{
	std::string msg = what;
	if(g_store.ctx) msg += std::string(": ") + abw_last_error(g_store.ctx);
	throw std::runtime_error(msg);
}

// the library's view of the root table, through public interfaces of the reference only
void build_table(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& root, HipStore& st, ClusterSeparatorHip::Table& t)
{
	const size_t N = scaf_db.ndps(), D = root.ndimensions(), S = scaf_db.nscafs();
	if(root.ndps() != N)
		throw std::invalid_argument("ClusterSeparatorHip: the first node must hold every datapoint (root of the recursion)");
	st.scaf_db = &scaf_db; st.N = N; st.D = D; st.S = S;
	st.values.assign(D, std::vector<double>());
	st.dim_names.assign(D, "");
	st.raw_of.assign(N + 1, 0);
	t.owner = &scaf_db; t.N = N; t.D = D; t.S = S;

	// keys: rank of every value among the column's distinct values (order-preserving, exact for any
	// .lrn; for abawaca-build's 3-decimal output the table has <= 1001 entries per signature column)
	t.keys.assign(D * N, 0);
	{
		// the dimensions are independent and only read here: one thread each, as many at a time as the reference's own
		// fan-out allows itself (reference src/abawaca.cpp:326-332 caps -p at 40)
		unsigned nt = std::max(1u, std::min(40u, std::thread::hardware_concurrency()));
		auto work = [&](unsigned first) {
			for(size_t d = first; d < D; d += nt) {
				const Dimension& dim = root.get_dimension(d + 1);
				st.dim_names[d] = dim.get_dimension_name();
				std::vector<double>& u = st.values[d];
				u.assign(dim.get_values().begin(), dim.get_values().end());	// std::set<double>: sorted, distinct
				for(auto it = dim.begin(); it != dim.end(); it++) {
					const size_t key = std::lower_bound(u.begin(), u.end(), it->value) - u.begin();
					t.keys[d * N + (it->dp - 1)] = (int32_t)key;
				}
			}
		};
		std::vector<std::thread> pool;
		for(unsigned k = 1; k < nt; k++) pool.emplace_back(work, k);
		work(0);
		for(auto& th : pool) th.join();
	}
	t.dp2scaf.resize(N);
	st.first_dp.assign(S + 1, 0);
	st.cnt.assign(S + 1, 0);
	for(size_t dp = N; dp >= 1; dp--) {
		const size_t sc = scaf_db.dp2scaf(dp);
		t.dp2scaf[dp - 1] = (int32_t)sc - 1;
		st.first_dp[sc] = dp;
	}
	t.scaf_len.resize(S);
	for(size_t s = 1; s <= S; s++) t.scaf_len[s - 1] = (int64_t)scaf_db.get_scaf(s)->get_seq_size();
	// scaffold -> SCG ids through SCGdb's public interface
	std::map<std::string, int32_t> scg_id;
	t.scg_off.assign(S + 1, 0);
	t.scg_ids.clear();
	for(size_t s = 1; s <= S; s++) {
		std::set<size_t> one; one.insert(s);
		std::map<std::string, size_t> found;
		scg_db.get_scg2count(one, found);
		for(auto it = found.begin(); it != found.end(); it++) {
			auto ins = scg_id.insert(std::pair<std::string, int32_t>(it->first, (int32_t)scg_id.size()));
			t.scg_ids.push_back(ins.first->second);
		}
		t.scg_off[s] = (int32_t)t.scg_ids.size();
	}
	t.n_scg = (int32_t)scg_id.size();
}

}	// namespace

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
ClusterSeparatorHip::ClusterSeparatorHip(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore) :
	ClusterSeparator(scaf_db, scg_db, clustering_db, threads_semaphore, new Result())
{}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorHip::loaded_for(const ScafDpData* db) const
{
	return g_store.ctx != NULL && g_store.scaf_db == db;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
// One resident column store on device 0 (reference src/abawaca.cpp:86 loads the table once, too)
void ClusterSeparatorHip::load_table(const Table& t)
{
	HipStore& st = g_store;
	if(!st.ctx && abw_ctx_create(abw_env_device(), &st.ctx) != ABW_OK)
		throw std::runtime_error("ClusterSeparatorHip: no MI355X / HIP device available");
	if(abw_sep_load(st.ctx, t.keys.data(), (int32_t)t.D, (int64_t)t.N, t.dp2scaf.data(), t.scaf_len.data(), (int32_t)t.S,
			t.n_scg ? t.scg_off.data() : NULL, t.n_scg ? t.scg_ids.data() : NULL, t.n_scg) != ABW_OK)
		fail("ClusterSeparatorHip: abw_sep_load failed");
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHip::run_node(const std::vector<int32_t>& scaf_ids, size_t n_dps, NodeResult& r)
{
	abw_split out;
	r.side.assign(scaf_ids.size(), 0);
	r.raw.assign(n_dps, 0);
	if(abw_sep_node(g_store.ctx, scaf_ids.data(), (int32_t)scaf_ids.size(), (int32_t)this->cluster_ndps_threshold, &out, NULL,
			r.side.data(), r.raw.data()) != ABW_OK)
		fail("ClusterSeparatorHip: abw_sep_node failed");
	r.dimension = out.dimension; r.key = out.key; r.legal = out.legal; r.split = out.split;
	r.sensitivity = out.sensitivity; r.specificity = out.specificity;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorHip::run_tree(std::vector<TreeRec>& nodes, std::vector<int32_t>& scaf2cluster)
{
	HipStore& st = g_store;
	std::vector<abw_tree_node> buf(1024);
	scaf2cluster.assign(st.S, 0);
	int64_t n = 0;
	int rc = abw_sep_tree(st.ctx, (int32_t)this->cluster_ndps_threshold, 0, buf.data(), (int64_t)buf.size(), &n, scaf2cluster.data());
	if(rc == ABW_ERR_CAPACITY && n > (int64_t)buf.size()) {
		buf.resize((size_t)n);
		rc = abw_sep_tree(st.ctx, (int32_t)this->cluster_ndps_threshold, 0, buf.data(), (int64_t)buf.size(), &n, scaf2cluster.data());
	}
	if(rc != ABW_OK) fail("ClusterSeparatorHip: abw_sep_tree failed");
	nodes.resize((size_t)n);
	for(int64_t i = 0; i < n; i++) {
		const abw_tree_node& b = buf[(size_t)i];
		TreeRec& r = nodes[(size_t)i];
		r.id = b.id; r.split = b.split; r.dimension = b.dimension; r.key = b.key; r.legal = b.legal; r.child1 = b.child1; r.child2 = b.child2;
		r.n_scaf = b.n_scaf; r.ndps = (long)b.ndps; r.sensitivity = b.sensitivity; r.specificity = b.specificity;
	}
	return true;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHip::run_assign(const std::vector<int32_t>& scaf_ids, size_t n_dps, int dimension, int key, NodeResult& r)
{
	abw_split out;
	r.side.assign(scaf_ids.size(), 0);
	r.raw.assign(n_dps, 0);
	if(abw_sep_node_assign(g_store.ctx, scaf_ids.data(), (int32_t)scaf_ids.size(), dimension, key, (int32_t)this->cluster_ndps_threshold, &out,
			r.side.data(), r.raw.data()) != ABW_OK)
		fail("ClusterSeparatorHip: abw_sep_node_assign failed");
	r.split = out.split;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHip::stats(size_t& from_tree, size_t& node_by_node)
{
	from_tree = g_store.from_tree; node_by_node = g_store.node_by_node;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterSeparatorHip::shutdown()
{
	if(g_store.ctx) abw_ctx_destroy(g_store.ctx);
	g_store = HipStore();
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
ostream& operator << (ostream& os, ClusterSeparatorHip& cs)
{
	// same text as reference src/ClusterSeparatorBySensitivitySpecificity.cpp:79-88
	os << " Best separation:\t" << "dimension " << cs.get_separating_dimension() << " ("
		<< cs.clustering_db.get_dimension(cs.get_separating_dimension()).get_dimension_name() << ")\t"
		<< "Value " << cs.get_separating_value() << "\t"
		<< "Specificity " << cs.get_best_specificity() << "\t"
		<< "Sensitivity " << cs.get_best_sensitivity();
	return os;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorHip::separate()
{
	// reference src/ClusterSeparator.cpp:59-67
	if(this->cluster1 != NULL) delete(this->cluster1);
	if(this->cluster2 != NULL) delete(this->cluster2);
	this->cluster1 = this->cluster2 = NULL;
	this->best_separation->reset();
	this->raw_dps_cluster1.clear();
	this->raw_dps_cluster2.clear();

	HipStore& st = g_store;
	if(!loaded_for(&scaf_db) || st.scaf_db != &scaf_db) {
		Table t;
		build_table(scaf_db, scg_db, clustering_db, st, t);
		load_table(t);
		// the root: the whole recursion in one call; the records serve every later separate() on this table
		st.tree.clear(); st.cursor = 0; st.from_tree = st.node_by_node = 0;
		if(!getenv("ABW_NODE_BY_NODE") && run_tree(st.tree, st.s2c)) {
			int max_id = 0;
			for(size_t i = 0; i < st.tree.size(); i++) max_id = std::max(max_id, std::max(st.tree[i].id, std::max(st.tree[i].child1, st.tree[i].child2)));
			st.idx_of.assign((size_t)max_id + 1, -1);
			for(size_t i = 0; i < st.tree.size(); i++) st.idx_of[st.tree[i].id] = (int32_t)i;
			// depth-first intervals: node X lies above bin b iff tin[X] <= tin[b] and tout[b] <= tout[X]
			st.tin.assign(st.tree.size(), 0); st.tout.assign(st.tree.size(), 0);
			std::vector<std::pair<int32_t, int> > stack;
			int32_t clock = 0;
			if(!st.tree.empty()) stack.push_back(std::pair<int32_t, int>(0, 0));
			while(!stack.empty()) {
				const int32_t i = stack.back().first;
				const int stage = stack.back().second++;
				const TreeRec& r = st.tree[i];
				if(stage == 0) st.tin[i] = clock++;
				const int child = stage == 0 ? r.child1 : stage == 1 ? r.child2 : 0;
				if(stage >= 2 || !r.split) { st.tout[i] = clock++; stack.pop_back(); continue; }
				if(child > 0 && child < (int)st.idx_of.size() && st.idx_of[child] >= 0) stack.push_back(std::pair<int32_t, int>(st.idx_of[child], 0));
			}
		}
	}

	// the node as a list of whole scaffolds (separate() always hands whole scaffolds to its children,
	// reference src/ClusterSeparator.cpp:94-122)
	const unordered_set<size_t>& dps = clustering_db.datapoints();
	std::vector<int32_t> scaf_ids;
	for(auto it = dps.begin(); it != dps.end(); it++) {
		const size_t sc = scaf_db.dp2scaf(*it);
		if(st.cnt[sc]++ == 0) scaf_ids.push_back((int32_t)sc - 1);
	}
	std::sort(scaf_ids.begin(), scaf_ids.end());
	const int32_t m = (int32_t)scaf_ids.size();
	bool whole = true;
	for(int32_t i = 0; i < m; i++) {
		const size_t sc = (size_t)scaf_ids[i] + 1;
		if(st.cnt[sc] != scaf_db.ndps(sc)) whole = false;
		st.cnt[sc] = 0;
	}
	if(!whole)
		throw std::invalid_argument("ClusterSeparatorHip: the node holds only part of a scaffold");

	// is this the node the recursion's records expect next?
	const TreeRec* rec = NULL;
	if(st.cursor < st.tree.size()) {
		const TreeRec& c = st.tree[st.cursor];
		bool same = (size_t)c.ndps == dps.size() && c.n_scaf == m;
		for(int32_t i = 0; same && i < m; i++) {
			const int32_t bin = st.s2c[scaf_ids[i]];
			const int32_t b = (bin > 0 && bin < (int32_t)st.idx_of.size()) ? st.idx_of[bin] : -1;
			same = b >= 0 && st.tin[st.cursor] <= st.tin[b] && st.tout[b] <= st.tout[st.cursor];
		}
		if(same) rec = &c;
		else st.tree.clear();					// another order of visits, or other nodes: node by node from here on
	}
	st.cursor++;

	NodeResult out;
	(rec ? st.from_tree : st.node_by_node)++;
	if(rec) {
		out.dimension = rec->dimension; out.key = rec->key; out.legal = rec->legal; out.split = rec->split;
		out.sensitivity = rec->sensitivity; out.specificity = rec->specificity;
		if(rec->split) {
			run_assign(scaf_ids, dps.size(), rec->dimension, rec->key, out);
			if(!out.split) throw std::logic_error("ClusterSeparatorHip: abw_sep_node_assign and abw_sep_tree disagree about a node");
		}
	}
	else
		run_node(scaf_ids, dps.size(), out);
	const std::vector<uint8_t>& side = out.side;
	const std::vector<uint8_t>& raw = out.raw;

	Result* best = static_cast<Result*>(best_separation);
	if(out.dimension > 0)
		best->set((size_t)out.dimension, st.values[out.dimension - 1][out.key], out.sensitivity, out.specificity, out.legal != 0);
	if(!out.legal || out.dimension <= 0)
		return false;								// reference src/ClusterSeparator.cpp:83-84

	// dps are listed to the library scaffold by scaffold, ascending dp id inside a scaffold -- consecutive ids from the
	// scaffold's first (reference src/ScafDpData.cpp:91-99)
	std::vector<uint8_t>& raw_of = st.raw_of;					// by datapoint id; only the node's entries are read
	{
		size_t off = 0;
		for(int32_t i = 0; i < m; i++) {
			const size_t sc = (size_t)scaf_ids[i] + 1, n_s = scaf_db.ndps(sc);
			for(size_t j = 0; j < n_s; j++) raw_of[st.first_dp[sc] + j] = raw[off + j];
			off += n_s;
		}
	}

	// The two clusters are built by the reference's own sequence of set operations, so that their unordered_sets
	// iterate in the reference's order (the cluster files are written in that order, reference
	// src/ClusterWriter.cpp:92-93).  Raw clusters (reference src/ClusterSeparator.cpp:36-47): the separating
	// Dimension is walked in the order of the node's datapoint set (reference src/Dimension.cpp:25-26).
	this->cluster1 = new Cluster(scaf_db);
	this->cluster2 = new Cluster(scaf_db);
	for(auto it = dps.begin(); it != dps.end(); it++)
		*(raw_of[*it] == 1 ? this->cluster1 : this->cluster2) += *it;
	for(auto sit = cluster1->dps_begin(); sit != cluster1->dps_end(); sit++) raw_dps_cluster1.insert(*sit);	// reference :87-90
	for(auto sit = cluster2->dps_begin(); sit != cluster2->dps_end(); sit++) raw_dps_cluster2.insert(*sit);

	if(!out.split) {								// reference :125-132
		delete(this->cluster1);
		delete(this->cluster2);
		this->cluster1 = this->cluster2 = NULL;
		best->reset();
		return false;
	}
	for(int32_t i = 0; i < m; i++)							// reference :94-101, decided by the library
		(side[i] == 1 ? this->cluster1 : this->cluster2)->add_assigned_scaf((size_t)scaf_ids[i] + 1);
	std::set<size_t> moved;								// reference :104-122
	for(auto sit = cluster1->dps_begin(); sit != cluster1->dps_end(); sit++)
		if(!cluster1->assigned(scaf_db.dp2scaf(*sit))) moved.insert(*sit);
	for(auto sit = moved.begin(); sit != moved.end(); sit++) { *cluster1 -= *sit; *cluster2 += *sit; }
	moved.clear();
	for(auto sit = cluster2->dps_begin(); sit != cluster2->dps_end(); sit++)
		if(!cluster2->assigned(scaf_db.dp2scaf(*sit))) moved.insert(*sit);
	for(auto sit = moved.begin(); sit != moved.end(); sit++) { *cluster1 += *sit; *cluster2 -= *sit; }
	return true;
}
