"""Host-side mirror of the reference's `ProposalNetwork` package (the 1000-cube proposal-and-scoring method)
over the fused geometry kernels of libcr3dod.so.  Same module / function names as the reference:
ProposalNetwork.utils.spaces.Cubes, .utils.conversions.cubes_to_box, .proposals.proposals.propose,
.scoring.scorefunction.score_*, .utils.plane.Plane."""
