// ClusterSeparatorKMeans — the reference-side binding of libabawaca_hip.so's k-means restarts.
//
// The reference has no k-means separator; this is the "third subclass at
// src/abawaca.cpp:109" the north-star asks for: a subclass of the reference's abstract ClusterSeparator
// (reference src/ClusterSeparator.h:15-75) with the constructor argument list of
// ClusterSeparatorBySensitivitySpecificity (reference ...Specificity.h:48-50).  It splits a node by the
// best of R randomly seeded k = 2 runs over the node's datapoints (all dimensions, z-scored over the
// root), then applies the reference's own post-processing of a raw split: the smaller raw side becomes
// cluster1 (reference src/ClusterSeparator.cpp:49-53), a scaffold follows the side holding at least half
// of its datapoints (:94-101), datapoints follow their scaffold (:103-122), both children must hold
// cluster_ndps_threshold datapoints (:125), and ClusterQuality::is_split_better (the SCG gate the
// reference's separator consults, reference src/ClusterQuality.cpp:96-136) must agree.
// It includes the reference's headers; it contains none of the reference's code.
#ifndef CLUSTERSEPARATORKMEANS_H
#define CLUSTERSEPARATORKMEANS_H

#include "ClusterSeparator.h"

class ClusterSeparatorKMeans : public ClusterSeparator {
protected:
	class Result : public ClusterSeparator::ClusteringResult {
	public:
		Result() : ClusterSeparator::ClusteringResult(), agreement(0), inertia(0), legal(false)	{}
		void	reset()						{agreement = inertia = 0; legal = false; ClusterSeparator::ClusteringResult::reset();}
		// dimension 0 = "all dimensions"; value = index of the winning restart
		void	set(size_t restart, double agree, double inert, bool l)	{dimension = 0; value = (double)restart; agreement = agree; inertia = inert; legal = l;}
		bool	is_better(const ClusterSeparator::ClusteringResult&) const	{return false;}	// restarts are compared in the library (inertia)
		bool	is_legal() const				{return legal;}
		void	print_separation_params(ostream& os) const	{os << "k-means restart: " << value << ", inertia: " << inertia << ", scaffold agreement: " << agreement;}
		double	agreement, inertia;
		bool	legal;
	};
public:
	ClusterSeparatorKMeans(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore);
	bool		separate() override;
	// fraction of the node's datapoints whose raw side is the side of their scaffold: plays the part of
	// both sensitivity and specificity of the reference's separator (same 0.8 acceptance level, ref ...Specificity.h:42-45)
	double		get_best_sensitivity() const		{return static_cast<Result*>(best_separation)->agreement;}
	double		get_best_specificity() const		{return static_cast<Result*>(best_separation)->agreement;}
	double		get_best_inertia() const		{return static_cast<Result*>(best_separation)->inertia;}
	friend ostream&	operator << (ostream& os, ClusterSeparatorKMeans& cs);
	// defaults 16, 20, 20240, -1.  chunk = points per fp32 partial sum (part of the result's bits): -1 = a rule that depends
	// on the node's size only (the same bits on any device), 0 = sized by the library for the device, else a multiple of 32
	static void	set_parameters(int restarts, int iterations, unsigned base_seed, int chunk = -1);
	static void	shutdown();
protected:
	void		separate_dimension(size_t) override	{}	// separate() is replaced as a whole
};

ostream& operator << (ostream& os, ClusterSeparatorKMeans& cs);

#endif // CLUSTERSEPARATORKMEANS_H
