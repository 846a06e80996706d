#!/usr/bin/env python3
"""python run.py -c config/<task>.json -p train|test   (the reference's command line; the driver is mdie_amd.host.run)"""
from mdie_amd.host import cli

if __name__ == "__main__":
    cli()
