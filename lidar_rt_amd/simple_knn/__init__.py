"""MI355X counterpart of the reference's `simple_knn` package (submodules/simple-knn)."""
