#!/usr/bin/env python3
"""Same entry point name as the reference's code/berkeley_hydro_main.py; see hydromodel_amd/cli.py."""
from hydromodel_amd.cli import run_cli

if __name__ == "__main__":
    run_cli()
