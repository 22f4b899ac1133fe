"""The block jig of this repository (what tests/jigs.lua is to the reference).

Importing it has the effect `package.loaded['tests.jigs']` has in the reference (radio/blocks/signal/firfilter.lua:57): a FIRFilterBlock
whose use_fft argument was not given runs the direct form, so the golden-vector tests see one output per input and the bits of the fmaf
chain unless a test asks for an FFT form explicitly.  tests/conftest.py sets the same flag for pytest runs; helper scripts that run in a
child process import `tests` for it."""
import luaradio_amd.block as _block

_block.TESTS_JIGS_LOADED = True
