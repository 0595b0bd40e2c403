#!/usr/bin/env python3
"""Placeholder; replaced below."""
