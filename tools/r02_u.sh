#!/bin/bash
export SIS3D_T16_STAGGER=0
timeout 300 python tools/t16_trace.py mask 2 2>&1 | grep -v -i warn | tail -9
timeout 300 python tools/t16_trace.py rpn 2 2>&1 | grep -v -i warn | tail -9
timeout 300 python tools/t16_trace.py rpn 0 2>&1 | grep -v -i warn | tail -9
