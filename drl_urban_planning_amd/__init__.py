"""Importable alias of the ``drl-urban-planning_amd/`` package directory.

The product package directory carries the reference repo's name (with a hyphen, which Python
cannot import); this stub makes ``import drl_urban_planning_amd`` resolve to it.  It holds no
code of its own.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'drl-urban-planning_amd')
__path__ = [_real]
_init = _os.path.join(_real, '__init__.py')
with open(_init) as _f:
    exec(compile(_f.read(), _init, 'exec'))
del _f, _init
