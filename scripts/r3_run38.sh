#!/bin/bash
bash scripts/refresh_profiles.sh r3
