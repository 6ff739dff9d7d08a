"""``gecco-hip``: GECCO's own command line with the HIP engine injected.

GECCO threads a CRF *class* through every sub-command (``gecco/cli/commands/__init__.py:127-137,160-163``:
``main(argv=None, console=None, *, program=..., crf_type=None, classifier_type=None, ...)``); this module is the
five-line console entry point INTEGRATION.md describes.  ``gecco-hip run --genome X.fna -o out`` then behaves like
``gecco run`` with every contig of a call scored in batched HIP launches.  GECCO itself is not a dependency of this
package (its ORF finder, HMMER wrapper and type classifier are reused, not rebuilt): without it the entry point says so.
"""
import sys
from typing import List, Optional


def main(argv: Optional[List[str]] = None) -> int:
    try:
        import gecco.cli  # the reference package, installed separately
    except ImportError as err:
        sys.stderr.write(
            "gecco-hip needs GECCO itself (`pip install gecco-tool`): it injects gecco_amd.crf.ClusterCRF into "
            f"gecco.cli.main(crf_type=...).  Import failed: {err}\n"
            "The table front end works without it: python -m gecco_amd.predict --genes ... --features ... -o OUT\n")
        return 2
    from .crf import ClusterCRF

    return int(gecco.cli.main(argv, crf_type=ClusterCRF) or 0)


if __name__ == "__main__":
    raise SystemExit(main())
