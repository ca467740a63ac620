"""GraphCast (operational 0.25 degree, 13 levels) on MI355X: icosahedral multi-mesh, typed graph network, HIP kernels."""
