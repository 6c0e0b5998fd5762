for st in 0 12 24 40; do echo "=== STAGGER $st"; ADAIN_W4_STAGGER=$st python tools/wino4_persist_probe.py 2>/dev/null | grep "==\|whole tile\|CUs with"; done
