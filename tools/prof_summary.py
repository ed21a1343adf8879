import sqlite3, sys
c=sqlite3.connect(sys.argv[1]); steps=float(sys.argv[2]) if len(sys.argv)>2 else 1
rows=c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
tot=sum(r[2] for r in rows)
print(f"total {tot/1e3/steps:.2f} ms/step over {steps} steps")
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 30]:
    print(f"{r[2]/1e3/steps:7.3f} ms/step {r[1]/steps:7.1f} calls {r[3]:8.1f} us  {r[4]:5.1f}%  {r[0][:100]}")
