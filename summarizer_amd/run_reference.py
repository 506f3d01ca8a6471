"""Run one of the reference's own entry scripts on the HIP path, with zero edits to the reference checkout:

    python -m summarizer_amd.run_reference /path/to/Summarizer/summarizer/main.py -m vasnet -s tvsum -c yes --local 16

installs the module aliases (`summarizer_amd.install_as_reference`), puts the checkout on sys.path exactly like
`main.py:5` does itself, and executes the script as `__main__` with the remaining arguments (main.py:75-103 parses them)."""
import os
import runpy
import sys

from . import install_as_reference


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or not os.path.isfile(argv[0]):
        raise SystemExit("usage: python -m summarizer_amd.run_reference <reference script, e.g. summarizer/main.py> [its arguments]")
    script = os.path.abspath(argv[0])
    sys.path.append(os.path.dirname(os.path.dirname(script)))        # the directory holding the `summarizer` package
    install_as_reference()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
