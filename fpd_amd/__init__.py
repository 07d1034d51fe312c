"""Import alias: the product package lives in `fast-human-pose-estimation.pytorch_amd/` (the name the
project layout prescribes, which is not a valid Python identifier); `import fpd_amd` resolves to it."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'fast-human-pose-estimation.pytorch_amd')
__path__ = [_REAL]
with open(_os.path.join(_REAL, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, '__init__.py'), 'exec'))
