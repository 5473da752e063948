"""Drop-in module name for DAS3R: `from simple_knn._C import distCUDA2` (/root/reference/scene/gaussian_model.py:21)."""
