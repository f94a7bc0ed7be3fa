"""Import alias: the package directory is ``de-i2i-gan_amd/`` (not a valid Python identifier), so this stub
package points its ``__path__`` there.  ``import de_i2i_gan_amd`` == the code under ``de-i2i-gan_amd/``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "de-i2i-gan_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
