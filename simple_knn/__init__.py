"""Import-name shim: ``from simple_knn._C import distCUDA2`` resolves to the MI355X-native implementation when the
repository root is on ``sys.path``."""
from youreditableavatar_amd.simple_knn import _C  # noqa: F401
