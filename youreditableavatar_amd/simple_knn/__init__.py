"""Drop-in ``simple_knn`` for MI355X: ``from simple_knn._C import distCUDA2`` (the package the reference installs from
Edit_core/thirdparties/simple-knn; ext.cpp:15-17 exports exactly this one function)."""
from . import _C  # noqa: F401
