"""Import alias for the package directory
`efficient-rag-with-learned-retrieval-and-uncertainty-quantification_amd/` (its name is not a valid
Python identifier).  `import rag_uq_amd` == that package; sub-modules resolve inside it."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "efficient-rag-with-learned-retrieval-and-uncertainty-quantification_amd")
__path__.insert(0, _REAL)
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
