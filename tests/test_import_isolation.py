"""The oracle's stub-import of the reference must not leak its ``lightning`` test double into the product package
(``pytest tests/test_oracle_vs_reference.py tests/test_checkpoint_keys.py`` failed in that order)."""
import subprocess
import sys

import pytest

from oracle import refimport


@pytest.mark.skipif(not refimport.available(), reason="/root/reference only exists in the build container")
def test_reference_import_then_product_import_in_a_fresh_interpreter():
    code = (
        "import sys\n"
        "from oracle import refimport\n"
        "ns = refimport.import_reference()\n"
        "assert not getattr(sys.modules.get('lightning'), '_oracle_stub', False), 'stub lightning left in sys.modules'\n"
        "import cultionet_amd.lightning as L\n"
        "assert hasattr(L.CultionetLitModel, 'load_from_checkpoint')\n"
        "assert ns.CultionetLitModel is not L.CultionetLitModel\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
