"""Put this directory on sys.path (before the reference's own build) and `import redner`
resolves to the MI355X implementation: `from redner_amd.redner import *`."""
from redner_amd.redner import *          # noqa: F401,F403
from redner_amd.redner import float_ptr, int_ptr   # noqa: F401
