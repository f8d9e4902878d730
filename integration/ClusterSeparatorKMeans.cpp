// See ClusterSeparatorKMeans.h.  Host-side adapter between the reference's data model and
// abw_km_load / abw_km_node (include/abawaca_hip.h): the feature table goes to the GPU once, at the root;
// every node of the recursion then sends only its scaffold list.
#include "ClusterSeparatorKMeans.h"
#include "abawaca_hip.h"
#include <stdlib.h>
#include <string.h>
// the GPU this process works on: ABW_DEVICE in the environment (default 0)
static inline int abw_env_device() { const char* e = getenv("ABW_DEVICE"); return e ? atoi(e) : 0; }
#include <algorithm>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct KmStore {
	abw_ctx*		ctx = NULL;
	const ScafDpData*	scaf_db = NULL;
	size_t			N = 0, D = 0;
	std::vector<float>	x;		// [D][N] z-scored over the root, dp id - 1 is the column (staging for abw_km_load)
	int			restarts = 16, iterations = 20;
	unsigned		base_seed = 20240;
	int			chunk = -1;	// points per fp32 partial sum: -1 = km_default_chunk (device-independent), 0 = the library's choice for the device
};
KmStore g_km;

// The chunk bounds the fp32 partial sums and so is part of the result's bits.  abw_km_node(chunk = 0) sizes it for the
// device it runs on; the adapter instead passes the size that rule gives on a 256-CU device (MI355X) WHATEVER the device
// is, so that a run is reproducible bit for bit across devices.  abw_km_split::chunk reports what was used.
int32_t km_default_chunk(int64_t n, int R)
{
	return abw_km_chunk_for(256, n, (int32_t)R);				// the library's own rule (no copy of it here), for 256 compute units
}

void km_fail(const std::string& what)
{
	std::string msg = what;
	if(g_km.ctx) msg += std::string(": ") + abw_last_error(g_km.ctx);
	throw std::runtime_error(msg);
}

// features of every datapoint, once: the root ClusterData through its public interface
// (ClusterData::get_dimension -> Dimension::begin/end, reference src/Dimension.h:33-37)
void km_load(const ScafDpData& scaf_db, const ClusterData& root)
{
	KmStore& st = g_km;
	if(!st.ctx && abw_ctx_create(abw_env_device(), &st.ctx) != ABW_OK)
		throw std::runtime_error("ClusterSeparatorKMeans: no MI355X / HIP device available");
	// ABW_KM_ARITH=bf16x3: the Lloyd passes on three bf16 planes (include/abawaca_hip.h, abw_km_set_arith); it takes effect
	// from 33 restarts on (set_parameters) -- with the default 16 there is one restart tile and the exact kernel runs
	if(const char* e = getenv("ABW_KM_ARITH"))
		if(abw_km_set_arith(st.ctx, strcmp(e, "bf16x3") == 0 ? ABW_KM_ARITH_BF16X3 : ABW_KM_ARITH_F32) != ABW_OK)
			throw std::runtime_error(abw_last_error(st.ctx));
	const size_t N = scaf_db.ndps(), D = root.ndimensions();
	if(root.ndps() != N)
		throw std::invalid_argument("ClusterSeparatorKMeans: the first node must hold every datapoint (root of the recursion)");
	if(D > 223)
		throw std::invalid_argument("ClusterSeparatorKMeans: more than 223 dimensions");
	st.scaf_db = &scaf_db; st.N = N; st.D = D;
	st.x.assign(D * N, 0.0f);
	for(size_t d = 0; d < D; d++) {
		const Dimension& dim = root.get_dimension(d + 1);
		double sum = 0, sq = 0;
		for(auto it = dim.begin(); it != dim.end(); it++) { sum += it->value; sq += it->value * it->value; }
		const double mean = sum / (double)N;
		const double var = std::max(0.0, sq / (double)N - mean * mean);
		const double inv = 1.0 / std::max(1e-6, std::sqrt(var));
		for(auto it = dim.begin(); it != dim.end(); it++)
			st.x[d * N + (it->dp - 1)] = (float)((it->value - mean) * inv);
	}
	std::vector<int32_t> dp2scaf(N);
	for(size_t dp = 1; dp <= N; dp++) dp2scaf[dp - 1] = (int32_t)scaf_db.dp2scaf(dp) - 1;
	if(abw_km_load(st.ctx, st.x.data(), (int32_t)D, (int64_t)N, dp2scaf.data(), (int32_t)scaf_db.nscafs()) != ABW_OK)
		km_fail("ClusterSeparatorKMeans: abw_km_load failed");
	std::vector<float>().swap(st.x);						// resident now
}

}	// namespace

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
ClusterSeparatorKMeans::ClusterSeparatorKMeans(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore) :
	ClusterSeparator(scaf_db, scg_db, clustering_db, threads_semaphore, new Result())
{}

void ClusterSeparatorKMeans::set_parameters(int restarts, int iterations, unsigned base_seed, int chunk)
{
	g_km.restarts = restarts; g_km.iterations = iterations; g_km.base_seed = base_seed; g_km.chunk = chunk;
}

void ClusterSeparatorKMeans::shutdown()
{
	if(g_km.ctx) abw_ctx_destroy(g_km.ctx);
	g_km = KmStore();
}

ostream& operator << (ostream& os, ClusterSeparatorKMeans& cs)
{
	os << " Best separation:\t";
	cs.print_separation_params(os);
	return os;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorKMeans::separate()
{
	// reference src/ClusterSeparator.cpp:59-67
	if(this->cluster1 != NULL) delete(this->cluster1);
	if(this->cluster2 != NULL) delete(this->cluster2);
	this->cluster1 = this->cluster2 = NULL;
	this->best_separation->reset();
	this->raw_dps_cluster1.clear();
	this->raw_dps_cluster2.clear();

	if(g_km.ctx == NULL || g_km.scaf_db != &scaf_db)
		km_load(scaf_db, clustering_db);
	KmStore& st = g_km;

	// the node as a list of whole scaffolds in ascending id order, which is ascending datapoint order too
	// (separate() always hands whole scaffolds to its children, reference src/ClusterSeparator.cpp:94-122)
	const unordered_set<size_t>& node_dps = clustering_db.datapoints();
	std::map<size_t, std::vector<size_t> > per_scaf;
	for(auto it = node_dps.begin(); it != node_dps.end(); it++) per_scaf[scaf_db.dp2scaf(*it)].push_back(*it);
	std::vector<int32_t> scaf_ids;
	for(auto it = per_scaf.begin(); it != per_scaf.end(); it++) {
		if(it->second.size() != scaf_db.ndps(it->first))
			throw std::invalid_argument("ClusterSeparatorKMeans: node holds only part of scaffold " + scaf_db.scaf_id2name(it->first));
		std::sort(it->second.begin(), it->second.end());
		scaf_ids.push_back((int32_t)it->first - 1);
	}
	const int32_t m = (int32_t)scaf_ids.size();
	const size_t n = node_dps.size();
	abw_km_split out;
	std::vector<uint8_t> side(m), raw(n);
	const int32_t chunk = st.chunk >= 0 ? st.chunk : km_default_chunk((int64_t)n, st.restarts);
	if(abw_km_node(st.ctx, scaf_ids.data(), m, st.restarts, st.base_seed, st.iterations, chunk, (int32_t)this->cluster_ndps_threshold,
			&out, side.data(), raw.data()) != ABW_OK)
		km_fail("ClusterSeparatorKMeans: abw_km_node failed");
	if(out.best_restart < 0 || out.raw1 < (int64_t)this->cluster_ndps_threshold)
		return false;								// too small, or no significant separation (reference :83-84)

	// raw sets (reference :87-90) and the two children (reference :94-122), as the library assigned them
	this->cluster1 = new Cluster(scaf_db);
	this->cluster2 = new Cluster(scaf_db);
	size_t k = 0;
	int32_t i = 0;
	for(auto it = per_scaf.begin(); it != per_scaf.end(); it++, i++) {
		Cluster* c = (side[i] == 1) ? this->cluster1 : this->cluster2;
		c->add_assigned_scaf(it->first);
		for(size_t j = 0; j < it->second.size(); j++, k++) {
			(raw[k] == 1 ? raw_dps_cluster1 : raw_dps_cluster2).insert(it->second[j]);
			*c += it->second[j];
		}
	}
	const bool legal = out.split && out.agreement >= 0.8 && cq.is_split_better(*this->cluster1, *this->cluster2);
	static_cast<Result*>(best_separation)->set((size_t)out.best_restart, out.agreement, out.inertia, legal);
	if(!legal) {									// reference :125-132
		delete(this->cluster1); delete(this->cluster2);
		this->cluster1 = this->cluster2 = NULL;
		return false;
	}
	return true;
}
