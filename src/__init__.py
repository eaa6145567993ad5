"""Alias package: the reference's import paths (`src.utils.registry`, `src.models`, ...) resolved to crdr_amd so
that plug-ins written against iwa-shi/CRDR register into this framework unchanged."""
