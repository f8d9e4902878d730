// ClusterSeparatorHip — the reference-side binding of libabawaca_hip.so's separator.
//
// This is the file a maintainer of CK7/abawaca adds to src/ (see INTEGRATION.md): a subclass of the
// reference's abstract ClusterSeparator (reference src/ClusterSeparator.h:15-75) with the constructor
// argument list of ClusterSeparatorBySensitivitySpecificity (reference
// src/ClusterSeparatorBySensitivitySpecificity.h:48-50), so that the only change in main is the type
// name at reference src/abawaca.cpp:109.  It includes the reference's headers; it contains none of
// the reference's code.
//
// The reference's driver asks node by node (src/abawaca.cpp:98-112), in ascending id order with the children of a split
// taking the next two ids (:155-194) -- the order in which abw_sep_tree returns its nodes.  The first separate() of a table
// (the root) therefore runs the WHOLE recursion in that one library call and keeps the node records; every later separate()
// finds its node in them (checked: datapoints, scaffolds, and every scaffold's final bin lying under that node), and only
// asks the library for the node's raw sets and scaffold sides (abw_sep_node_assign) to rebuild the reference's Cluster
// objects.  A driver that visits nodes in another order, or other nodes, gets abw_sep_node node by node, as does
// ABW_NODE_BY_NODE=1 in the environment.
#ifndef CLUSTERSEPARATORHIP_H
#define CLUSTERSEPARATORHIP_H

#include "ClusterSeparator.h"
#include <stdint.h>
#include <vector>

class ClusterSeparatorHip : public ClusterSeparator {
protected:
	// same fields and rules as ClusterSeparatorBySensitivitySpecificity::ClusteringResult
	// (reference ...Specificity.h:13-46); the rules themselves are evaluated on the GPU / in the library.
	class Result : public ClusterSeparator::ClusteringResult {
	public:
		Result() : ClusterSeparator::ClusteringResult(), sensitivity(0), specificity(0), legal(false)	{}
		void	reset()						{sensitivity = specificity = 0; legal = false; ClusterSeparator::ClusteringResult::reset();}
		void	set(size_t d, double v, double sens, double spec, bool l)	{dimension = d; value = v; sensitivity = sens; specificity = spec; legal = l;}
		bool	is_better(const ClusterSeparator::ClusteringResult&) const	{return false;}	// comparisons happen in the library
		bool	is_legal() const				{return legal;}
		void	print_separation_params(ostream& os) const	{ClusterSeparator::ClusteringResult::print_separation_params(os); os << ", Specificity: " << specificity << ", sensitivity:" << sensitivity;}
		double	sensitivity, specificity;
		bool	legal;
	};
public:
	ClusterSeparatorHip(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore);
	bool		separate() override;
	double		get_best_sensitivity() const		{return static_cast<Result*>(best_separation)->sensitivity;}
	double		get_best_specificity() const		{return static_cast<Result*>(best_separation)->specificity;}
	friend ostream&	operator << (ostream& os, ClusterSeparatorHip& cs);
	// Releases the GPU context and the resident column store (optional; call once at program end).
	static void	shutdown();
	// separate() calls answered from the one-call recursion's records / by abw_sep_node, since the table was loaded
	static void	stats(size_t& from_tree, size_t& node_by_node);
	// The two places where the table meets the GPU; ClusterSeparatorHipSharded spreads them over several contexts.
	// Everything is in the numbering of the library: scaffold ids and datapoint positions 0-based, datapoints scaffold by
	// scaffold.  The table itself (keys, scaffold layout, SCGs) is built once from the root node's ClusterData.
	struct Table {
		const ScafDpData*		owner;		// the ScafDpData the table describes
		size_t				N, D, S;
		std::vector<int32_t>		keys;		// [D][N] rank of every value among its column's distinct values
		std::vector<int32_t>		dp2scaf;	// [N]
		std::vector<int64_t>		scaf_len;	// [S]
		std::vector<int32_t>		scg_off, scg_ids;	// CSR scaffold -> SCG ids
		int32_t				n_scg;
	};
	struct TreeRec { int id, split, dimension, key, legal, child1, child2, n_scaf; long ndps; double sensitivity, specificity; };	// abw_tree_node
	struct NodeResult {					// what abw_sep_node returns (include/abawaca_hip.h)
		int		dimension, key, legal, split;
		double		sensitivity, specificity;
		std::vector<uint8_t> side;			// [m] 1 / 2 per scaffold of the node, in the order of scaf_ids
		std::vector<uint8_t> raw;			// [n] raw side per datapoint, scaffold by scaffold, ascending datapoint id
	};
protected:
	void		separate_dimension(size_t) override	{}	// separate() is replaced as a whole
	// hooks of separate(): the resident table, one node
	virtual void	load_table(const Table& t);
	virtual void	run_node(const std::vector<int32_t>& scaf_ids, size_t n_dps, NodeResult& r);
	virtual bool	loaded_for(const ScafDpData* db) const;
	// the whole recursion in one call: node records in id order and every scaffold's final bin; false = not available here
	virtual bool	run_tree(std::vector<TreeRec>& nodes, std::vector<int32_t>& scaf2cluster);
	// raw sets and scaffold sides of a node whose winner is known (from run_tree's records)
	virtual void	run_assign(const std::vector<int32_t>& scaf_ids, size_t n_dps, int dimension, int key, NodeResult& r);
};

ostream& operator << (ostream& os, ClusterSeparatorHip& cs);

#endif // CLUSTERSEPARATORHIP_H
