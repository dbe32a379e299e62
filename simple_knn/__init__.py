"""Top-level alias so that the reference's import line `from simple_knn._C import distCUDA2`
(lib/scene/gaussian_model.py:16) resolves to the MI355X implementation."""
