"""Import-name alias for the reference's `simple_knn` package (volume_rendering/gaussian_model.py:25)."""
